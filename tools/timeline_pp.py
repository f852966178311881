#!/usr/bin/env python3
"""Debug: s_memtime timeline of wave 0 (group P) and wave 4 (group Q) of workgroup 0 of the ping-pong edge update
(needs tools/libexp_TLP.so = the engine built with -DMGN_TIMELINE).  Prints, per group, the mean duration in shader
cycles of: the wait at the barrier that opens a compute slot, the GEMM quarter, the wait at the barrier that opens the
free slot, the free slot's work, its closing part (fragment pre-load, DMA issue, counted drain) -- by position (unit,
quarter) within a tile.
usage: python tools/timeline_pp.py [save]"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["MGN_LIB"] = os.path.join(R, "tools", os.environ.get("TL_LIB", "libexp_TLP.so"))
os.environ["MGN_PP"] = "2"
import torch, graph_physics_amd as gp
from graph_physics_amd import ops, _capi
dev = torch.device("cuda:0")
g = gp.cylinder_batch(int(os.environ.get("TL_BATCH", "16")), 1885, 0).to(dev)
topo = ops.Topology(g.edge_index, g.x.shape[0])
N, E, H = topo.N, topo.E, 128
f = dict(dtype=torch.float32, device=dev)
x, e = torch.randn(N, H, **f), torch.randn(E, H, **f)
W0 = torch.randn(H, 3 * H, **f) * 0.05
Wh = [torch.randn(H, H, **f) * 0.09 for _ in range(3)]
bs = [torch.zeros(H, **f) for _ in range(4)]
sc = torch.ones(H, **f)
Pd, Ps = x @ W0[:, H:2 * H].t(), x @ W0[:, 2 * H:].t()
e_new = torch.empty(E, H, **f)
save = len(sys.argv) > 1 and sys.argv[1] == "save"
He = [torch.empty(E, H, **f) for _ in range(3)] if save else None
Ue, Re = (torch.empty(E, H, **f), torch.empty(E, **f)) if save else (None, None)
Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)] if save else None
agg, part = torch.empty(N, H, **f), torch.empty((E + 15) // 16, 2, H, **f)
pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(W0.data_ptr(), 3 * H, False, units[0])] + [(Wh[l].data_ptr(), H, False, units[l + 1]) for l in range(3)], dev)
L = _capi.lib()
L.mgn_debug_pp_timeline.restype = C.c_int
buf = (C.c_ulonglong * (2 * 1024))()
n = (C.c_int * 2)()
for it in range(3):
    ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, None, He, Ue, Re, ldw0=3 * H,
                adds=[(Pd, topo.dst_s), (Ps, topo.src_s)], wpk=units, saveM=Me, seg=(topo.dst_s, topo.rowptr_dst, agg, part))
    torch.cuda.synchronize()
    L.mgn_debug_pp_timeline(buf, n)
names = {(0, 1): "bar->C", (1, 2): "C (gemm)", (2, 3): "bar->F", (4, 0): "F close"}
for grp in range(2):
    st = [(buf[grp * 1024 + i] >> 8, int(buf[grp * 1024 + i] & 255)) for i in range(n[grp])]
    nsteps = sum(1 for _, t in st if t == 0)
    print(f"group {'PQ'[grp]}: {n[grp]} stamps = {nsteps} steps; whole stream {st[-1][0] - st[0][0]} cycles, {(st[-1][0] - st[0][0]) / max(nsteps, 1):.0f} per step")
    # intervals keyed by (position in the tile, previous tag, tag)
    acc, cnt = {}, {}
    step = -1
    for k in range(len(st) - 1):
        if st[k][1] == 0:
            step += 1
        key = (step % 16, st[k][1], st[k + 1][1])
        acc[key] = acc.get(key, 0) + st[k + 1][0] - st[k][0]
        cnt[key] = cnt.get(key, 0) + 1
    for p in range(16):
        keys = [k for k in acc if k[0] == p]
        keys.sort(key=lambda k: [0, 1, 2, 3, 10, 11, 12, 20, 21, 22, 23, 24, 4].index(k[1]))
        print(f"   (u{p // 4},q{p % 4}) " + "  ".join(f"{names.get((k[1], k[2]), str(k[1]) + '>' + str(k[2]))} {acc[k] / cnt[k]:5.0f}" for k in keys))
