#!/usr/bin/env python3
"""Debug: s_memtime timeline of wave 0 (h = 0) and wave 4 (h = 1) of workgroup 0 of the register-resident-weights edge update
(csrc/mgn_ppr.inc; needs tools/libexp_TLR.so = the engine built with -DMGN_TIMELINE: tools/mkvar.sh TLR -DMGN_TIMELINE).
Per half, mean shader cycles of the eight intervals of a group: wait at the barrier that opens the first unit, the first
unit's matrix phase (+ its closing wait / row sums), wait at the mid barrier, the boundary phase, and the same for the second unit.
usage: python tools/timeline_ppr.py [save]"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["MGN_LIB"] = os.path.join(R, "tools", os.environ.get("TL_LIB", "libexp_TLR.so"))
os.environ["MGN_PPR"] = "2"
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
save = len(sys.argv) > 1 and sys.argv[1] in ("save", "bwd")
bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
He = [torch.empty(E, H, **f) for _ in range(3)] if save else None
Ue, Re = (torch.empty(E, H, **f), torch.empty(E, **f)) if save else (None, None)
Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)] if save else None
agg, part = torch.empty(N, H, **f), torch.empty((E + 15) // 16, 2, H, **f)
pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(W0.data_ptr(), 3 * H, False, units[0])] + [(Wh[l].data_ptr(), H, False, units[l + 1]) for l in range(3)], dev)
L = _capi.lib()
L.mgn_debug_ppr_timeline.restype = C.c_int
buf = (C.c_ulonglong * (2 * 160))()
n = (C.c_int * 2)()
if bwd:
    bu = [pk.data_ptr() for _ in range(4)]
    pkb = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
    bu = [pkb.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
    ops.wpack([(Wh[2].data_ptr(), H, True, bu[0]), (Wh[1].data_ptr(), H, True, bu[1]), (Wh[0].data_ptr(), H, True, bu[2]), (W0.data_ptr(), 3 * H, True, bu[3])], dev)
    ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, None, He, Ue, Re, ldw0=3 * H,
                adds=[(Pd, topo.dst_s), (Ps, topo.src_s)], wpk=units, saveM=Me, seg=(topo.dst_s, topo.rowptr_dst, agg, part))
    dOut, dAgg = torch.randn(E, H, **f), torch.randn(N, H, **f)
    dZ = [torch.empty(E, H, **f) for _ in range(4)]
    dE_o, dsc = torch.empty(E, H, **f), torch.empty(H, **f)
reps = int(os.environ.get("TL_REPS", "3"))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(reps):
    if it == reps - 1:
        ev0.record()
    if bwd:
        ops.mlp_bwd(E, H, 4, dOut, dAgg, topo.dst_s, H, Ue, Re, sc, He, [None] * 4, dZ, [(None, dOut, dE_o)], [None] * 4, dsc, wpk=bu, Ms=Me)
    else:
        ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, None, He, Ue, Re, ldw0=3 * H,
                    adds=[(Pd, topo.dst_s), (Ps, topo.src_s)], wpk=units, saveM=Me, seg=(topo.dst_s, topo.rowptr_dst, agg, part))
ev1.record()
torch.cuda.synchronize()
L.mgn_debug_ppr_timeline(buf, n)
print(f"E = {E}, save = {save}, bwd = {bwd}, last launch {ev0.elapsed_time(ev1) * 1e3:.1f} us (with the timeline stamps)")
names = ["bar->M0", "M0", "bar->B0", "B0", "bar->M1", "M1", "bar->B1", "B1"]
for hh in range(2):
    st = [(buf[hh * 160 + i] >> 8, int(buf[hh * 160 + i] & 255)) for i in range(n[hh])]
    ngrp = sum(1 for _, t in st if t == 0)
    tot = st[-1][0] - st[0][0]
    print(f"half h={hh}: {n[hh]} stamps = {ngrp} groups; loop {tot} cycles, {tot / max(ngrp, 1):.0f} per group (2 tiles x 4 units: floor 2 x 96 x 16 x 2 = 6144)")
    acc = [0] * 8
    cnt = [0] * 8
    main = [x for x in st if x[1] < 8]
    for k in range(len(main) - 1):
        t = main[k][1]
        acc[t] += main[k + 1][0] - main[k][0]
        cnt[t] += 1
    print("   " + "  ".join(f"{names[t]} {acc[t] / max(cnt[t], 1):5.0f}" for t in range(8)))
    # inside the boundary phases: intervals between consecutive stamps, keyed by (phase tag, previous tag, tag)
    fine, fc = {}, {}
    ph = None
    for k in range(len(st) - 1):
        if st[k][1] < 8:
            ph = st[k][1]
        if ph in (3, 7) and (st[k + 1][1] >= 8 or st[k][1] >= 8):
            key = (ph, st[k][1], st[k + 1][1])
            fine[key] = fine.get(key, 0) + st[k + 1][0] - st[k][0]
            fc[key] = fc.get(key, 0) + 1
    tagn = {3: "B0 start", 7: "B1 start", 8: "tile0 relu/saves", 9: "tile1 relu/saves", 10: "selector", 11: "piece 1", 12: "piece 2", 13: "piece 3", 14: "epilogue tile 0", 15: "epilogue tile 1", 4: "end", 0: "end"}
    for key in sorted(fine):
        print(f"      {names[key[0]]}: {tagn.get(key[1], key[1])} -> {tagn.get(key[2], key[2])}: {fine[key] / fc[key]:6.0f}")
