#!/usr/bin/env python3
"""Debug: s_memrealtime timeline of wave 0 of the first / last workgroups of the split-bf16
edge-forward kernel (needs tools/libexp_TLX.so = the engine built with -DMGN_TIMELINE)."""
import ctypes as C, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["MGN_LIB"] = os.path.join(R, "tools", os.environ.get("TL_LIB", "libexp_TLX.so"))  # build: hipcc ... -DMGN_TIMELINE -o tools/libexp_TLX.so csrc/*.hip
try:
    import torch, graph_physics_amd as gp
    from graph_physics_amd import ops, _capi
    dev = torch.device("cuda:0")
    nb = int(os.environ.get("TL_BATCH", "16"))  # meshes in the batch (1: the single-tile latency regime)
    PREC = int(os.environ.get("TL_PREC", "0"))   # 1: the one-term (bf16 matrix mode) instance, 2: with two-byte saves
    g = gp.cylinder_batch(nb, 1885, 0).to(dev)
    topo = ops.Topology(g.edge_index, g.x.shape[0])
    N, E, H = topo.N, topo.E, 128
    f = dict(dtype=torch.float32, device=dev)
    x, e = torch.randn(N, H, **f), torch.randn(E, H, **f)
    W0 = torch.randn(H, 3 * H, **f) * 0.05
    Wh = [torch.randn(H, H, **f) * 0.09 for _ in range(3)]
    bs = [torch.zeros(H, **f) for _ in range(4)]
    sc = torch.ones(H, **f)
    Pd, Ps = x @ W0[:, H:2 * H].t(), x @ W0[:, 2 * H:].t()
    m, e_new = torch.empty(E, H, **f), torch.empty(E, H, **f)
    save = len(sys.argv) > 1 and sys.argv[1] == "save"
    node = len(sys.argv) > 1 and sys.argv[1] == "node"
    He = [torch.empty(E, H, **f) for _ in range(3)] if save else None
    Ue, Re = (torch.empty(E, H, **f), torch.empty(E, **f)) if save else (None, None)
    Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)] if save else None
    agg, part = torch.empty(N, H, **f), torch.empty((E + 15) // 16, 2, H, **f)   # production shape: aggregation fused, no message store
    pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
    units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
    ops.wpack([(W0.data_ptr(), 3 * H, False, units[0])] + [(Wh[l].data_ptr(), H, False, units[l + 1]) for l in range(3)], dev)
    L = _capi.lib()
    L.mgn_debug_timeline.restype = C.c_int
    buf = (C.c_ulonglong * (8 * 512))(); pos = (C.c_int * 8)()
    if node:  # node update shape: N rows, phases (x, agg), 4 layers, two post-products
        pkn = torch.empty(7 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
        un = [pkn.data_ptr() + u * _capi.WPACK_BYTES for u in range(7)]
        Wn0 = torch.randn(H, 2 * H, **f) * 0.05
        ops.wpack([(Wn0.data_ptr(), 2 * H, False, un[0]), (Wn0.data_ptr() + 4 * H, 2 * H, False, un[1])] +
                  [(Wh[l].data_ptr(), H, False, un[2 + l]) for l in range(3)] +
                  [(W0.data_ptr() + 4 * H, 3 * H, False, un[5]), (W0.data_ptr() + 8 * H, 3 * H, False, un[6])], dev)
        agg, x_new = torch.randn(N, H, **f), torch.empty(N, H, **f)
        Hn = [torch.empty(N, H, **f) for _ in range(3)]
        Un, Rn = torch.empty(N, H, **f), torch.empty(N, **f)
        Pdn, Psn = torch.empty(N, H, **f), torch.empty(N, H, **f)
    for it in range(3):
        if node:
            ops.mlp_fwd(N, H, [(x, None, H), (agg, None, H)], [Wn0] + Wh, bs, sc, H, x, x_new, None, Hn, Un, Rn,
                        posts=[(W0.data_ptr() + 4 * H, Pdn), (W0.data_ptr() + 8 * H, Psn)], post_ldw=3 * H, wpk=un, precision=PREC)
        else:
            ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, None, He, Ue, Re, ldw0=3 * H,
                        adds=[(Pd, topo.dst_s), (Ps, topo.src_s)], wpk=units, saveM=Me, seg=(topo.dst_s, topo.rowptr_dst, agg, part),
                        precision=PREC)
        torch.cuda.synchronize()
        L.mgn_debug_timeline(buf, pos)
    if hasattr(L, "mgn_debug_census"):
        cen = (C.c_ulonglong * (1024 * 4))()
        L.mgn_debug_census(cen)
        nwg = min(512, (E + 63) // 64)
        t0 = min(cen[4 * b + 2] for b in range(nwg))
        place = {}
        for b in range(nwg):
            hw, xcc = cen[4 * b], cen[4 * b + 1] & 15
            key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)   # XCC, SE, SH, CU
            place.setdefault(key, []).append((b, (hw >> 4) & 3, cen[4 * b + 2] - t0, cen[4 * b + 3] - t0))
        print("placement census: %d distinct CUs hold the %d workgroups" % (len(place), nwg))
        import collections
        print("  workgroups per CU:", dict(collections.Counter(len(v) for v in place.values())))
        diffs = collections.Counter()
        for key, v in sorted(place.items())[:12]:
            print("  CU", key, "-> (wg, simd of wave 0, start, end):", v)
        for v in place.values():
            if len(v) == 2:
                diffs[abs(v[0][0] - v[1][0])] += 1
        print("  blockIdx distance of co-resident pairs:", dict(diffs.most_common(8)))
        sd = [abs(v[0][2] - v[1][2]) for v in place.values() if len(v) == 2]
        if sd:
            sd.sort(); print("  start-time distance of pairs (10 ns ticks): median %d, p10 %d, p90 %d" % (sd[len(sd) // 2], sd[len(sd) // 10], sd[9 * len(sd) // 10]))
    names = {1: "T", 3: "pre", 4: "bar", 5: "gemm", 6: "drain", 7: "epi", 8: "end", 9: "norm+st", 10: "nbk"}
    firsts = [(buf[b * 512] >> 8) for b in range(8)]
    lasts = [(buf[b * 512 + max(pos[b] - 1, 0)] >> 8) for b in range(8)]
    print("census (100 MHz ticks = 10 ns): first stamp rel. to WG0:", [f - firsts[0] for f in firsts])
    print("                                last stamp rel. to WG0 first:", [l - firsts[0] for l in lasts])
    for b in (0, 1, 4):
        n = pos[b]
        ev = [(buf[b * 512 + i] >> 8, buf[b * 512 + i] & 255) for i in range(n)]
        print(f"--- workgroup slot {b}: {n} stamps; deltas in 10 ns ticks (tag = interval ENDING at that stamp)")
        tot = {}
        line = []
        for i in range(1, n):
            d = ev[i][0] - ev[i - 1][0]
            tag = names[ev[i][1]]
            tot[tag] = tot.get(tag, 0) + d
            line.append(f"{tag}{d}")
            if ev[i][1] == 8:
                print("  " + " ".join(line)); line = []
        print("  totals:", tot, "span", ev[n - 1][0] - ev[0][0])
finally:
    pass
