#!/usr/bin/env python3
"""Debug: s_memtime timeline of wave 0 (producer of A) and wave 4 (consumer) of workgroup 0 of k_wgrad_pc (needs
tools/libexp_TLW.so = the engine built with -DMGN_TIMELINE).  Producer stamps: 0 loop top (after the barrier), 1 rows arrived
(+ column sums), 2 split + piece writes issued, 3 next loads issued; consumer: 0 after the barrier, 1 MFMAs + reads issued.
usage: python tools/timeline_wpc.py"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["MGN_LIB"] = os.path.join(R, "tools", os.environ.get("TL_LIB", "libexp_TLW.so"))
import torch, graph_physics_amd as gp
from graph_physics_amd import ops, _capi
dev = torch.device("cuda:0")
g = gp.cylinder_batch(16, 1885, 0).to(dev)
E, H = g.edge_index.shape[1], 128
f = dict(dtype=torch.float32, device=dev)
dZ = [torch.randn(E, H, **f) for _ in range(4)]
X = [torch.randn(E, H, **f) for _ in range(4)]
gW = [torch.empty(H, H, **f) for _ in range(4)]
gb = [torch.empty(H, **f) for _ in range(4)]
L = _capi.lib()
L.mgn_debug_wpc_timeline.restype = C.c_int
buf = (C.c_ulonglong * (2 * 2048))()
n = (C.c_int * 2)()
reps = int(os.environ.get("TL_REPS", "3"))   # TL_REPS=60: the timeline of the LAST of 60 back-to-back launches (the sustained regime)
for it in range(reps):
    ops.wgrad([(dZ[l], H, 8, X[l], H, 8, H, gW[l], 0, H, gb[l]) for l in range(4)], dev)
    if reps <= 3:
        torch.cuda.synchronize()
torch.cuda.synchronize()
L.mgn_debug_wpc_timeline(buf, n)
for grp, name in ((0, "producer"), (1, "consumer")):
    st = [(buf[grp * 2048 + i] >> 8, int(buf[grp * 2048 + i] & 255)) for i in range(n[grp])]
    rt = [c for c, t in st if t == 8]
    cyc = [c for c, t in st if t in (7, 9)]
    st = [x for x in st if x[1] != 8]
    if len(rt) == 2 and len(cyc) == 2:
        us = (rt[1] - rt[0]) / 100.0
        print(f"{name}: kernel body {us:.1f} us = {cyc[1] - cyc[0]} shader cycles -> {(cyc[1] - cyc[0]) / us / 1e3:.2f} GHz")
    nt = sum(1 for _, t in st if t == 0)
    print(f"{name}: {n[grp]} stamps, {nt} tiles, {(st[-1][0] - st[0][0]) / max(nt - 1, 1):.0f} cycles per tile")
    acc, cnt = {}, {}
    for k in range(len(st) - 1):
        key = (st[k][1], st[k + 1][1])
        acc[key] = acc.get(key, 0) + st[k + 1][0] - st[k][0]
        cnt[key] = cnt.get(key, 0) + 1
    for key in sorted(acc):
        print(f"   {key[0]} -> {key[1]}: {acc[key] / cnt[key]:7.0f} cycles  (x{cnt[key]})")
    # a few raw tiles from the middle
    mid = len(st) // 2
    print("   raw:", " ".join(f"{t}:{c - st[mid][0]}" for c, t in st[mid:mid + 14]))

# per-workgroup start / end (s_memrealtime, 100 MHz)
L.mgn_debug_wpc_wgtimes.restype = C.c_int
wb = (C.c_ulonglong * 1024)()
L.mgn_debug_wpc_wgtimes(wb)
import statistics
t_first = min(wb[2 * i] for i in range(256))
dur = [(wb[2 * i + 1] - wb[2 * i]) / 100.0 for i in range(256)]
start = [(wb[2 * i] - t_first) / 100.0 for i in range(256)]
end = [(wb[2 * i + 1] - t_first) / 100.0 for i in range(256)]
print(f"workgroups: duration min {min(dur):.1f} median {statistics.median(dur):.1f} max {max(dur):.1f} us; start spread {max(start):.1f} us; last end {max(end):.1f} us")
for x in range(8):
    d = [dur[i] for i in range(256) if i % 8 == x]
    print(f"   blockIdx % 8 == {x}: duration min {min(d):.1f} median {statistics.median(d):.1f} max {max(d):.1f}")
for jb in range(4):
    d = dur[64 * jb: 64 * jb + 64]
    print(f"   job {jb}: duration min {min(d):.1f} median {statistics.median(d):.1f} max {max(d):.1f}")
