#!/usr/bin/env python3
"""The fused edge backward (mgn_edge_bwd_fused) on the bench shape: HIP-event timing next to the split launches it
replaces (mgn_mlp_bwd edge chain + the four E-row weight-gradient jobs), results compared, and -- with a -DMGN_TIMELINE
build named by MGN_LIB -- the s_memrealtime phase timeline of wave 0 of the first workgroups.
usage: python tools/kbench_fused.py [batch] [timeline]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import graph_physics_amd as gp
from graph_physics_amd import ops, _capi
from tools.kbench import timeit

dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = gp.cylinder_batch(nb, 1885, 0).to(dev)
topo = ops.Topology(g.edge_index, g.x.shape[0])
N, E, H = topo.N, topo.E, 128
f = dict(dtype=torch.float32, device=dev)
torch.manual_seed(0)
x, e = torch.randn(N, H, **f), torch.randn(E, H, **f)
W0 = torch.randn(H, 3 * H, **f) * 0.05
Wh = [torch.randn(H, H, **f) * 0.09 for _ in range(3)]
bs = [torch.randn(H, **f) * 0.1 for _ in range(4)]
sc = torch.rand(H, **f) + 0.5
Pd, Ps = x @ W0[:, H:2 * H].t(), x @ W0[:, 2 * H:].t()
pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(W0.data_ptr(), 3 * H, False, units[0])] + [(Wh[l].data_ptr(), H, False, units[l + 1]) for l in range(3)], dev)
m, e_new = torch.empty(E, H, **f), torch.empty(E, H, **f)
He = [torch.empty(E, H, **f) for _ in range(3)]
Ue, Re = torch.empty(E, H, **f), torch.empty(E, **f)
Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)]
ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, m, He, Ue, Re, ldw0=3 * H, adds=[(Pd, topo.dst_s), (Ps, topo.src_s)],
            wpk=units, saveM=Me)
de, dagg = torch.randn(E, H, **f), torch.randn(N, H, **f)
pkb = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
ub = [pkb.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(Wh[2].data_ptr(), H, True, ub[0]), (Wh[1].data_ptr(), H, True, ub[1]), (Wh[0].data_ptr(), H, True, ub[2]), (W0.data_ptr(), 3 * H, True, ub[3])], dev)
nbk = H // 16


def split():
    dZ = [torch.empty(E, H, **f) for _ in range(4)]
    de_new, dsc = torch.empty(E, H, **f), torch.empty(H, **f)
    gW = [torch.empty(H, 3 * H, **f)] + [torch.empty(H, H, **f) for _ in range(3)]
    gb = [torch.empty(H, **f) for _ in range(4)]

    def fn():
        ops.mlp_bwd(E, H, 4, de, dagg, topo.dst_s, H, Ue, Re, sc, He, [None] * 4, dZ, [(None, de, de_new)], [None] * 4, dsc, wpk=ub, Ms=Me)
        ops.wgrad([(dZ[0], H, nbk, e, H, nbk, H, gW[0], 0, 3 * H, gb[0])] +
                  [(dZ[l], H, nbk, He[l - 1], H, nbk, H, gW[l], 0, H, gb[l]) for l in range(1, 4)], dev)
    return fn, (dZ[0], de_new, dsc, gW, gb)


def fused():
    dZ0, de_new, dsc = torch.empty(E, H, **f), torch.empty(E, H, **f), torch.empty(H, **f)
    gW = [torch.zeros(H, 3 * H, **f)] + [torch.empty(H, H, **f) for _ in range(3)]
    gb = [torch.empty(H, **f) for _ in range(4)]

    def fn():
        ops.edge_bwd_fused(E, de, dagg, topo.dst_s, Ue, Re, sc, [e] + He, Me, ub, de_new, dZ0,
                           [(gW[0], 0, 3 * H), (gW[1], 0, H), (gW[2], 0, H), (gW[3], 0, H)], gb, dsc)
    return fn, (dZ0, de_new, dsc, gW, gb)


fs, rs = split()
ff, rf = fused()
fs(); ff()
torch.cuda.synchronize()


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


print(f"N={N} E={E}")
print("fused vs split: dZ0 %.1e dE %.1e dscale %.1e" % (rel(rf[0], rs[0]), rel(rf[1], rs[1]), rel(rf[2], rs[2])),
      "dW", ["%.1e" % rel(rf[3][l][:, :H], rs[3][l][:, :H]) for l in range(4)], "db", ["%.1e" % rel(rf[4][l], rs[4][l]) for l in range(4)])
print(f"split (chain + 4 E-row weight-gradient jobs): {timeit(fs) * 1e3:8.1f} us")
print(f"fused                                       : {timeit(ff) * 1e3:8.1f} us")
if len(sys.argv) > 2 and sys.argv[2] == "timeline":
    L = _capi.lib()
    if not hasattr(L, "mgn_debug_timeline"):
        raise SystemExit("timeline: needs a -DMGN_TIMELINE build (MGN_LIB=...)")
    buf = (C.c_ulonglong * (8 * 512))()
    pos = (C.c_int * 8)()
    ff()
    torch.cuda.synchronize()
    L.mgn_debug_timeline(buf, pos)
    names = {1: "tile start", 2: "loop end", 3: "partials written"}
    for k in range(4):
        names.update({10 + k: f"L{3 - k} put+split", 20 + k: f"L{3 - k} chain unit", 30 + k: f"L{3 - k} drain+barrier", 40 + k: f"L{3 - k} wgrad"})
    for slot in (0, 1):
        n = pos[slot]
        ev = [(buf[slot * 512 + i] >> 8, int(buf[slot * 512 + i] & 255)) for i in range(n)]
        if not ev:
            continue
        tot = {}
        cnt = {}
        for (t0, _), (t1, tag) in zip(ev[:-1], ev[1:]):
            tot[tag] = tot.get(tag, 0) + (t1 - t0)
            cnt[tag] = cnt.get(tag, 0) + 1
        span = (ev[-1][0] - ev[0][0]) * 0.01
        ntile = cnt.get(10, 1)
        print(f"-- workgroup slot {slot}: {n} stamps, {span:.1f} us, {ntile} tiles; mean us per tile by phase (time spent reaching the stamp):")
        for tag in sorted(tot):
            print(f"   {names.get(tag, tag):>18s}: {tot[tag] * 0.01 / max(ntile, 1):7.2f} us/tile  (x{cnt[tag]})")
