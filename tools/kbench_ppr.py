#!/usr/bin/env python3
"""Register-resident-weights edge update (csrc/mgn_ppr.inc) against the x6 static-shape kernel and the ping-pong kernel on the bench
edge shape: accuracy of every output against fp64, bit equality of the saves with x6, HIP-event timing, inference and training mode.
usage: python tools/kbench_ppr.py [batch=16] [sizes: comma list of row counts for the ragged-tail check | none]"""
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
pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(W0.data_ptr(), 3 * H, False, units[0])] + [(Wh[l].data_ptr(), H, False, units[l + 1]) for l in range(3)], dev)


def run(M, save, pp):
    """the edge update on rows [0, M) with the aggregation fused; returns (outputs dict, launch closure)"""
    os.environ["MGN_PP"] = "2" if pp == 1 else "0"
    os.environ["MGN_PPR"] = "2" if pp == 2 else "0"
    sl = slice(0, M)
    nn = int(topo.dst_s[M - 1]) + 1  # destinations covered by the row range (rows are dst-sorted)
    rowptr = torch.searchsorted(topo.dst_s[sl].contiguous(), torch.arange(nn + 1, device=dev, dtype=torch.int32)).to(torch.int32)
    e_new = torch.full((M, H), float("nan"), **f)
    agg = torch.full((nn, H), float("nan"), **f)
    part = torch.full(((M + 15) // 16, 2, H), float("nan"), **f)
    He = [torch.full((M, H), float("nan"), **f) for _ in range(3)] if save else None
    Ue, Re = (torch.full((M, H), float("nan"), **f), torch.full((M,), float("nan"), **f)) if save else (None, None)
    Me = [torch.zeros(M, 4, dtype=torch.int32, device=dev) for _ in range(3)] if save else None
    seg = (topo.dst_s[sl].contiguous(), rowptr, agg, part)

    def fn():
        ops.mlp_fwd(M, H, [(e[sl], None, H)], [W0] + Wh, bs, sc, H, e[sl], e_new, None, He, Ue, Re, ldw0=3 * H,
                    adds=[(Pd, topo.dst_s[sl].contiguous()), (Ps, topo.src_s[sl].contiguous())], wpk=units, saveM=Me, seg=seg)

    def full():
        fn()
        ops.seg_fix(rowptr, part, agg)

    full()
    torch.cuda.synchronize()
    out = {"e_new": e_new, "agg": agg}
    if save:
        out.update({"H1": He[0], "H2": He[1], "H3": He[2], "U": Ue, "R": Re, "M1": Me[0], "M2": Me[1], "M3": Me[2]})
    return out, fn


# fp64 reference of the same update (all rows)
d = torch.float64
z = e.to(d) @ W0[:, :H].to(d).t() + Pd.to(d)[topo.dst_s.long()] + Ps.to(d)[topo.src_s.long()] + bs[0].to(d)
hs = []
for l in range(3):
    h = z.clamp_min(0)
    hs.append(h)
    z = h @ Wh[l].to(d).t() + bs[l + 1].to(d)
rms = z.norm(dim=1, keepdim=True) / H ** 0.5
u_ref = z / (rms + 1e-8)
m_ref = sc.to(d) * u_ref
e_ref = e.to(d) + m_ref


def rel(a, b):
    return float((a.to(d) - b).abs().max() / b.abs().max())


def ref_of(M):
    nn = int(topo.dst_s[M - 1]) + 1
    agg = torch.zeros(nn, H, dtype=d, device=dev).index_add_(0, topo.dst_s[:M].long(), m_ref[:M])
    r = {"e_new": e_ref[:M], "agg": agg, "H1": hs[0][:M], "H2": hs[1][:M], "H3": hs[2][:M], "U": u_ref[:M], "R": rms[:M, 0]}
    return r


sizes = ([] if sys.argv[2] == "none" else [int(s) for s in sys.argv[2].split(",")]) if len(sys.argv) > 2 else [E, E - 1, E - 77, 70001, 40000, 257, 128, 129, 33, 17, 1]
ok = True
for M in sizes:
    ref = ref_of(M)
    for save in (False, True):
        a, _ = run(M, save, 0)
        b, _ = run(M, save, 2)
        line = f"M={M:7d} save={int(save)}"
        for k in b:
            if k[0] in "HM" and not torch.equal(a[k], b[k]):
                line += f" {k}:NOT-BIT-EQUAL-TO-X6"
                ok = False
            if k.startswith("M"):  # sign bits must match the saved activation of the SAME run
                Hk = b["H" + k[1]]
                bits = (Hk.view(M, 8, 4, 4) > 0).permute(0, 2, 1, 3).reshape(M, 4, 32).long()
                want = (bits << torch.arange(32, device=dev)).sum(-1)
                good = torch.equal(b[k].long() & 0xffffffff, want)
                line += f" {k}:{'ok' if good else 'BAD'}"
                ok = ok and good
                continue
            ea, eb = rel(a[k], ref[k]), rel(b[k], ref[k])
            good = eb <= max(2e-6, 2 * ea) and not bool(torch.isnan(b[k]).any())
            line += f" {k}:{eb:.1e}({ea:.1e}){'' if good else '!!'}"
            ok = ok and good
        print(line, flush=True)
print("ACCURACY vs fp64 (ppr, x6 in brackets):", "OK" if ok else "FAILED")

def rotating(save, pp, K=6):
    """K sets of output tensors used in turn: every launch writes memory no earlier launch of the loop has left in a cache -- what a
    training step does (15 rounds, fresh saves each); the same buffers over and over flatter the write path by ~20 %"""
    fns = [run(E, save, pp)[1] for _ in range(K)]
    state = [0]

    def fn():
        fns[state[0] % K]()
        state[0] += 1
    return fn


for save in (False, True):
    for rep in range(2):
        for pp in (0, 1, 2):
            if pp == 1 and save:
                continue
            _, fn = run(E, save, pp)
            t = timeit(fn)
            if save:
                os.environ["MGN_PP"] = "2" if pp == 1 else "0"
                os.environ["MGN_PPR"] = "2" if pp == 2 else "0"
                tr = timeit(rotating(save, pp), iters=18)
                print(f"{('x6 static', 'ping-pong', 'ppr')[pp]:10s} save={int(save)} {tr*1e3:8.1f} us  with 6 rotating sets of outputs", flush=True)
                os.environ["MGN_PP"] = "2" if pp == 1 else "0"
                os.environ["MGN_PPR"] = "2" if pp == 2 else "0"
            print(f"{('x6 static', 'ping-pong', 'ppr')[pp]:10s} save={int(save)} {t*1e3:8.1f} us  {6 * 8.0*E*H*H/t/1e9:7.1f} TFLOP/s bf16", flush=True)
