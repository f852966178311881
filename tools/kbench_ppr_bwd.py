#!/usr/bin/env python3
"""Register-resident-weights edge BACKWARD chain (csrc/mgn_ppr.inc, k_edge_bwd_ppr) against the x6 static-shape kernel on the bench
edge shape: every output (dZ[0..3], dE, dscale) against an fp64 evaluation, and HIP-event timing of both.
usage: python tools/kbench_ppr_bwd.py [batch=16] [sizes: comma list of row counts | none]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import graph_physics_amd as gp
from graph_physics_amd import ops, _capi
from tools.kbench import timeit

dev = torch.device("cuda:0")
g = gp.cylinder_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1885, 0).to(dev)
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
dOut, dAgg = torch.randn(E, H, **f), torch.randn(N, H, **f)
pk = torch.empty(8 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
fu = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
bu = [pk.data_ptr() + (4 + u) * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(W0.data_ptr(), 3 * H, False, fu[0])] + [(Wh[l].data_ptr(), H, False, fu[l + 1]) for l in range(3)], dev)
ops.wpack([(Wh[2].data_ptr(), H, True, bu[0]), (Wh[1].data_ptr(), H, True, bu[1]), (Wh[0].data_ptr(), H, True, bu[2]), (W0.data_ptr(), 3 * H, True, bu[3])], dev)

# forward saves (the x6 kernel), all rows
os.environ["MGN_PPR"] = "0"
e_new, agg, part = torch.empty(E, H, **f), torch.empty(N, H, **f), torch.empty((E + 15) // 16, 2, H, **f)
He = [torch.empty(E, H, **f) for _ in range(3)]
Ue, Re = torch.empty(E, H, **f), torch.empty(E, **f)
Me = [torch.zeros(E, 4, dtype=torch.int32, device=dev) for _ in range(3)]
ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, None, He, Ue, Re, ldw0=3 * H,
            adds=[(Pd, topo.dst_s), (Ps, topo.src_s)], wpk=fu, saveM=Me, seg=(topo.dst_s, topo.rowptr_dst, agg, part))
torch.cuda.synchronize()

# fp64 backward from the SAVED forward state (masks from the saved activations: both kernels read the same bits)
d = torch.float64
dY = dOut.to(d) + dAgg.to(d)[topo.dst_s.long()]
U64, R64 = Ue.to(d), Re.to(d)
gg = sc.to(d) * dY
dz3 = gg / (R64[:, None] + 1e-8) - U64 * ((gg * U64).sum(1, keepdim=True) / (H * R64[:, None]))
dz2 = (dz3 @ Wh[2].to(d)) * (He[2] > 0)
dz1 = (dz2 @ Wh[1].to(d)) * (He[1] > 0)
dz0 = (dz1 @ Wh[0].to(d)) * (He[0] > 0)
dE = dOut.to(d) + dz0 @ W0[:, :H].to(d)


def run(M, ppr):
    os.environ["MGN_PPR"] = "2" if ppr else "0"
    sl = slice(0, M)
    dZ = [torch.full((M, H), float("nan"), **f) for _ in range(4)]
    dE_o = torch.full((M, H), float("nan"), **f)
    dsc = torch.full((H,), float("nan"), **f)
    dst = topo.dst_s[sl].contiguous()

    def fn():
        ops.mlp_bwd(M, H, 4, dOut[sl], dAgg, dst, H, Ue[sl], Re[sl], sc, [t[sl] for t in He], [None] * 4, dZ, [(None, dOut[sl], dE_o)], [None] * 4, dsc,
                    wpk=bu, Ms=[t[sl] for t in Me])

    fn()
    torch.cuda.synchronize()
    return {"dZ0": dZ[0], "dZ1": dZ[1], "dZ2": dZ[2], "dZ3": dZ[3], "dE": dE_o, "dscale": dsc}, fn


def rel(a, b):
    return float((a.to(d) - b).abs().max() / b.abs().max())


sizes = ([] if sys.argv[2] == "none" else [int(s) for s in sys.argv[2].split(",")]) if len(sys.argv) > 2 else [E, E - 1, E - 77, 70001, 40000, 257, 129, 128, 33, 17, 1]
ok = True
for M in sizes:
    ref = {"dZ0": dz0[:M], "dZ1": dz1[:M], "dZ2": dz2[:M], "dZ3": dz3[:M], "dE": dE[:M], "dscale": (dY[:M] * U64[:M]).sum(0)}
    a_, _ = run(M, False)
    b_, _ = run(M, True)
    line = f"M={M:7d}"
    for k in b_:
        ea, eb = rel(a_[k], ref[k]), rel(b_[k], ref[k])
        good = eb <= max(2e-6, 2 * ea) and not bool(torch.isnan(b_[k]).any())
        line += f" {k}:{eb:.1e}({ea:.1e}){'' if good else '!!'}"
        ok = ok and good
    print(line, flush=True)
print("ACCURACY vs fp64 (ppr, x6 in brackets):", "OK" if ok else "FAILED")
def rotating(ppr, K=6):
    fns = [run(E, ppr)[1] for _ in range(K)]
    state = [0]

    def fn():
        fns[state[0] % K]()
        state[0] += 1
    return fn


for rep in range(2):
    for ppr in (False, True):
        os.environ["MGN_PPR"] = "2" if ppr else "0"
        tr = timeit(rotating(ppr), iters=18)
        print(f"{'ppr' if ppr else 'x6 static':10s} bwd chain {tr*1e3:8.1f} us  with 6 rotating sets of outputs", flush=True)
        _, fn = run(E, ppr)
        t = timeit(fn)
        print(f"{'ppr' if ppr else 'x6 static':10s} bwd chain {t*1e3:8.1f} us  {6 * 8.0*E*H*H/t/1e9:7.1f} TFLOP/s bf16", flush=True)
