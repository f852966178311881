#!/usr/bin/env python3
"""Kernel micro-benchmarks on the bench workload shapes (HIP-event timing).
usage: python tools/kbench.py [--batch 16] [--what edge,node,seg,bwd,wgrad]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import graph_physics_amd as gp
from graph_physics_amd import ops


def timeit(fn, iters=20, warm=3, rounds=5):
    """min over `rounds` of the average launch time (run-to-run noise is ~+-8 %)"""
    for _ in range(warm):
        fn()
    best = None
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        b.synchronize()
        t = a.elapsed_time(b) / iters
        best = t if best is None else min(best, t)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--nodes", type=int, default=1885)
    ap.add_argument("--what", default="edge,edge_nosave,node,seg,bwd,wgrad")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = gp.cylinder_batch(a.batch, a.nodes, 0).to(dev)
    topo = ops.Topology(g.edge_index, g.x.shape[0])
    N, E, H = topo.N, topo.E, 128
    f = dict(dtype=torch.float32, device=dev)
    torch.manual_seed(0)
    x, e = torch.randn(N, H, **f), torch.randn(E, H, **f)
    W0 = torch.randn(H, 3 * H, **f) * 0.05
    Wn0 = torch.randn(H, 2 * H, **f) * 0.05
    Wh = [torch.randn(H, H, **f) * 0.09 for _ in range(3)]
    bs = [torch.zeros(H, **f) for _ in range(4)]
    sc = torch.ones(H, **f)
    m, e_new, agg, x_new = torch.empty(E, H, **f), torch.empty(E, H, **f), torch.empty(N, H, **f), torch.empty(N, H, **f)
    He = [torch.empty(E, H, **f) for _ in range(3)]
    Ue, Re = torch.empty(E, H, **f), torch.empty(E, **f)
    Hn = [torch.empty(N, H, **f) for _ in range(3)]
    Un, Rn = torch.empty(N, H, **f), torch.empty(N, **f)
    flops_e = 12.0 * E * H * H
    flops_n = 10.0 * N * H * H
    what = a.what.split(",")
    print(f"N={N} E={E}")

    def rep(name, ms, flops=None, byts=None):
        s = f"{name:14s} {ms*1e3:9.1f} us"
        if flops:
            s += f"  {flops/ms/1e9:7.1f} TFLOP/s ({flops/ms/1e9/157.3*100:4.1f}% of fp32 MFMA)"
        if byts:
            s += f"  {byts/ms/1e6:7.0f} GB/s"
        print(s, flush=True)

    ph_e = [(e, None, H), (x, topo.dst_s, H), (x, topo.src_s, H)]
    if "edge" in what:
        rep("edge fwd+save", timeit(lambda: ops.mlp_fwd(E, H, ph_e, [W0] + Wh, bs, sc, H, e, e_new, m, He, Ue, Re)), flops_e, 4.0 * E * H * 7)
    if "edge_nosave" in what:
        rep("edge fwd", timeit(lambda: ops.mlp_fwd(E, H, ph_e, [W0] + Wh, bs, sc, H, e, e_new, m)), flops_e, 4.0 * E * H * 3)
        rep("edge fwd 1ph", timeit(lambda: ops.mlp_fwd(E, H, [(e, None, H)], [W0[:, :H].contiguous()] + Wh, bs, sc, H, e, e_new, m)), 8.0 * E * H * H)
    if "node" in what:
        ph_n = [(x, None, H), (agg, None, H)]
        rep("node fwd+save", timeit(lambda: ops.mlp_fwd(N, H, ph_n, [Wn0] + Wh, bs, sc, H, x, x_new, None, Hn, Un, Rn)), flops_n)
    if "seg" in what:
        rep("segsum dst", timeit(lambda: ops.segsum(m, topo.rowptr_dst, None, agg), 50), None, 4.0 * E * H + 4.0 * N * H)
        rep("segsum src", timeit(lambda: ops.segsum(m, topo.rowptr_src, topo.perm_src, agg), 50), None, 4.0 * E * H + 4.0 * N * H)
    if "bwd" in what:
        ops.mlp_fwd(E, H, ph_e, [W0] + Wh, bs, sc, H, e, e_new, m, He, Ue, Re)
        dZ = [torch.empty(E, H, **f) for _ in range(4)]
        de, dagg, de_new = torch.randn(E, H, **f), torch.randn(N, H, **f), torch.empty(E, H, **f)
        WT = [None] + [w.t().contiguous() for w in Wh]
        WT0 = W0[:, :H].t().contiguous()
        db = [torch.empty(H, **f) for _ in range(4)]
        dsc = torch.empty(H, **f)
        rep("edge bwd chain", timeit(lambda: ops.mlp_bwd(E, H, 4, de, dagg, topo.dst_s, H, Ue, Re, sc, He, WT, dZ, [(WT0, de, de_new)], [None] * 4, dsc)), 8.0 * E * H * H)
    if "wgrad" in what:
        dZ = [torch.randn(E, H, **f) for _ in range(4)]
        Sd = torch.randn(N, H, **f)
        g0, gh = torch.empty(H, 3 * H, **f), [torch.empty(H, H, **f) for _ in range(3)]
        nb = H // 16
        jobs = [(dZ[0], H, nb, e, H, nb, H, g0, 0, 3 * H), (Sd, H, nb, x, H, nb, H, g0, H, 3 * H), (Sd, H, nb, x, H, nb, H, g0, 2 * H, 3 * H)]
        for l in range(3):
            jobs.append((dZ[l + 1], H, nb, He[l], H, nb, H, gh[l], 0, H))
        fl = 2.0 * H * H * (4 * E + 2 * N)
        rep("wgrad (edge)", timeit(lambda: ops.wgrad(jobs, dev)), fl, 4.0 * H * (8 * E + 4 * N))
        dbs = [torch.empty(H, **f) for _ in range(4)]
        jobs_db = [jobs[0] + (dbs[0],), jobs[1], jobs[2]] + [jobs[3 + l] + (dbs[1 + l],) for l in range(3)]
        rep("wgrad + db", timeit(lambda: ops.wgrad(jobs_db, dev)), fl, 4.0 * H * (8 * E + 4 * N))


if __name__ == "__main__":
    main()
