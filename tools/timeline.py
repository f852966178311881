#!/usr/bin/env python3
"""Debug: dump the s_memtime timeline of wave 0 of the first workgroups of the edge-forward
kernel (needs tools/libexp_TL.so = the engine built with -DMGN_TIMELINE)."""
import ctypes as C, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
lib_path = os.path.join(R, "graph-physics_amd", "csrc", "libmgn_hip.so")
shutil.copy(lib_path, "/tmp/libmgn_orig.so")
shutil.copy(os.path.join(R, "tools", "libexp_TL.so"), lib_path)
try:
    import torch, graph_physics_amd as gp
    from graph_physics_amd import ops, _capi
    dev = torch.device("cuda:0")
    g = gp.cylinder_batch(16, 1885, 0).to(dev)
    topo = ops.Topology(g.edge_index, g.x.shape[0])
    N, E, H = topo.N, topo.E, 128
    f = dict(dtype=torch.float32, device=dev)
    x, e = torch.randn(N, H, **f), torch.randn(E, H, **f)
    W0 = torch.randn(H, 3 * H, **f) * 0.05
    Wh = [torch.randn(H, H, **f) * 0.09 for _ in range(3)]
    bs = [torch.zeros(H, **f) for _ in range(4)]
    sc = torch.ones(H, **f)
    m, e_new = torch.empty(E, H, **f), torch.empty(E, H, **f)
    ph = [(e, None, H), (x, topo.dst_s, H), (x, topo.src_s, H)]
    L = _capi.lib()
    L.mgn_debug_timeline.restype = C.c_int
    buf = (C.c_ulonglong * (8 * 512))(); pos = (C.c_int * 8)()
    which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    dZ = [torch.randn(E, H, **f) for _ in range(4)]
    Hs = [torch.randn(E, H, **f) for _ in range(3)]
    g0, gh = torch.empty(H, 3 * H, **f), [torch.empty(H, H, **f) for _ in range(3)]
    nb = H // 16
    jobs = [(dZ[0], H, nb, e, H, nb, H, g0, 0, 3 * H)] + [(dZ[l + 1], H, nb, Hs[l], H, nb, H, gh[l], 0, H) for l in range(3)]
    for it in range(3):
        if which == "fwd":
            ops.mlp_fwd(E, H, ph, [W0] + Wh, bs, sc, H, e, e_new, m)
        else:
            ops.wgrad(jobs, dev)
        torch.cuda.synchronize()
        L.mgn_debug_timeline(buf, pos)
    names = {1: "tile_start", 2: "prework", 3: "pre_barrier", 4: "barrier", 5: "half0_issued", 6: "drained", 7: "gemm_done", 8: "epilogue"}
    firsts = [(buf[b * 512] >> 8) for b in range(8)]
    lasts = [(buf[b * 512 + max(pos[b] - 1, 0)] >> 8) for b in range(8)]
    print("census (s_memrealtime, 100 MHz ticks): first stamp rel. to WG0:", [f - firsts[0] for f in firsts])
    print("                                       last  stamp rel. to WG0 first:", [l - firsts[0] for l in lasts])
    for b in (0, 1):
        n = pos[b]
        ev = [(buf[b * 512 + i] >> 8, buf[b * 512 + i] & 255) for i in range(n)]
        print(f"--- workgroup {b}: {n} stamps; deltas in cycles")
        t0 = ev[0][0]
        line = []
        for i in range(1, min(n, 140)):
            line.append(f"{names[ev[i][1]]}+{ev[i][0] - ev[i-1][0]}")
            if ev[i][1] == 8 or (which != "fwd" and ev[i][1] == 7):
                print("  tile: " + " ".join(line)); line = []
        print("  total span", ev[min(n, 140) - 1][0] - t0)
finally:
    shutil.copy("/tmp/libmgn_orig.so", lib_path)
