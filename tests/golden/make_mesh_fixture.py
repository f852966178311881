#!/usr/bin/env python3
"""Mint tests/golden/cylinder_vtu_mesh.npz from the reference's own test mesh
(/root/reference/tests/mock_vtu/cylinder_0.vtu, the file its dataset tests read): node
positions and triangle connectivity as plain arrays -- data, not code.  The VTU is the
zlib-compressed base64 flavour meshio writes; decoded here by hand (meshio is not installed)."""
import base64, os, re, struct, sys, zlib
import numpy as np

SRC = "/root/reference/tests/mock_vtu/cylinder_0.vtu"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cylinder_vtu_mesh.npz")


def decode(b64: str, dtype):
    b64 = b64.strip()
    # header: [nblocks, blocksize, last_blocksize, csize_0, ...] as uint32, base64-encoded on its own
    head = base64.b64decode(b64[:16])  # 12 bytes -> first three words
    nblocks, = struct.unpack("<I", head[:4])
    hbytes = 4 * (3 + nblocks)
    hlen = ((hbytes + 2) // 3) * 4
    hdr = np.frombuffer(base64.b64decode(b64[:hlen])[:hbytes], dtype="<u4")
    data = base64.b64decode(b64[hlen:])
    out, off = b"", 0
    for cs in hdr[3:]:
        out += zlib.decompress(data[off:off + int(cs)])
        off += int(cs)
    return np.frombuffer(out, dtype=dtype)


xml = open(SRC).read()
npts = int(re.search(r'NumberOfPoints="(\d+)"', xml).group(1))
ncells = int(re.search(r'NumberOfCells="(\d+)"', xml).group(1))
arrays = {m.group(2): (m.group(1), m.group(3)) for m in
          re.finditer(r'<DataArray type="(\w+)" Name="(\w+)"[^>]*format="binary">\s*([^<]+)</DataArray>', xml)}
tmap = {"Float32": "<f4", "Float64": "<f8", "Int64": "<i8", "Int32": "<i4", "UInt8": "u1"}
pts = decode(arrays["Points"][1], tmap[arrays["Points"][0]]).reshape(npts, 3)
conn = decode(arrays["connectivity"][1], tmap[arrays["connectivity"][0]])
offs = decode(arrays["offsets"][1], tmap[arrays["offsets"][0]])
types = decode(arrays["types"][1], tmap[arrays["types"][0]])
assert (types == 5).all() and (np.diff(np.concatenate([[0], offs])) == 3).all(), "triangles only"
face = conn.reshape(ncells, 3).T.astype(np.int32)       # PyG layout [3, F]
assert np.abs(pts[:, 2]).max() == 0.0
np.savez_compressed(OUT, pos=pts[:, :2].astype(np.float32), face=face)
print(OUT, "N", npts, "F", ncells, os.path.getsize(OUT), "bytes")
