#!/usr/bin/env python3
"""Split-bf16 (x6) vs exact-fp32 MFMA forward kernels on the bench edge/node shapes: HIP-event
timing and error of both against an fp64 torch evaluation of the same MLP."""
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


def run(wpk, save):
    m, e_new = torch.empty(E, H, **f), torch.empty(E, H, **f)
    He = [torch.empty(E, H, **f) for _ in range(3)] if save else None
    Ue, Re = (torch.empty(E, H, **f), torch.empty(E, **f)) if save else (None, None)
    Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)] if (save and wpk) else None
    fn = lambda: ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, m, He, Ue, Re, ldw0=3 * H,
                             adds=[(Pd, topo.dst_s), (Ps, topo.src_s)], wpk=wpk, saveM=Me)
    fn()
    torch.cuda.synchronize()
    return m, e_new, He, Ue, fn


# fp64 reference of the same MLP
d = torch.float64
z = e.to(d) @ W0[:, :H].to(d).t() + Pd.to(d)[topo.dst_s.long()] + Ps.to(d)[topo.src_s.long()] + bs[0].to(d)
hs = []
for l in range(3):
    h = z.clamp_min(0)
    hs.append(h)
    z = h @ Wh[l].to(d).t() + bs[l + 1].to(d)
rms = z.norm(dim=1, keepdim=True) / H ** 0.5
m_ref = sc.to(d) * z / (rms + 1e-8)


def rel(a, b):
    return float((a.to(d) - b).abs().max() / b.abs().max())


print(f"N={N} E={E}")
for name, wpk in (("fp32 MFMA", ()), ("split-bf16 x6", units)):
    for save in (False, True):
        m, e_new, He, Ue, fn = run(wpk, save)
        t = timeit(fn)
        s = f"{name:14s} save={int(save)} {t*1e3:8.1f} us  {8.0*E*H*H/t/1e9:7.1f} TFLOP/s   err m {rel(m, m_ref):.2e}  e' {rel(e_new, e.to(d) + m_ref):.2e}"
        if save:
            s += f"  H3 {rel(He[2], hs[2]):.2e}"
        print(s, flush=True)

# ---- backward chain: fp32 vs x6 (timing; values checked against each other)
m, e_new, He, Ue, fn = run((), True)
Re = torch.empty(E, **f)
Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)]
ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, m, He, Ue, Re, ldw0=3 * H, adds=[(Pd, topo.dst_s), (Ps, topo.src_s)], wpk=units, saveM=Me)
for l in range(3):  # mask bits against the saved activations
    bits = (He[l].view(E, 8, 4, 4) > 0).permute(0, 2, 1, 3).reshape(E, 4, 32).long()  # [row][g][4*ib+r]
    want = (bits << torch.arange(32, device=dev)).sum(-1)
    got = Me[l].long() & 0xffffffff
    assert torch.equal(got, want), f"mask bits of layer {l + 1} differ"
print("mask bits == (saved activation > 0)")
de, dagg = torch.randn(E, H, **f), torch.randn(N, H, **f)
WT = [None] + [w.t().contiguous() for w in Wh]
WT0 = W0[:, :H].t().contiguous()
pkb = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
ub = [pkb.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(Wh[2].data_ptr(), H, True, ub[0]), (Wh[1].data_ptr(), H, True, ub[1]), (Wh[0].data_ptr(), H, True, ub[2]), (W0.data_ptr(), 3 * H, True, ub[3])], dev)
res = {}
for name, wpk in (("fp32 MFMA", ()), ("split-bf16 x6", ub)):
    dZ = [torch.empty(E, H, **f) for _ in range(4)]
    de_new, dsc = torch.empty(E, H, **f), torch.empty(H, **f)
    fnb = lambda: ops.mlp_bwd(E, H, 4, de, dagg, topo.dst_s, H, Ue, Re, sc, He, WT, dZ, [(WT0, de, de_new)], [None] * 4, dsc, wpk=wpk, Ms=Me)
    fnb()
    torch.cuda.synchronize()
    res[name] = (dZ, de_new, dsc)
    t = timeit(fnb)
    print(f"bwd chain {name:14s} {t*1e3:8.1f} us  {8.0*E*H*H/t/1e9:7.1f} TFLOP/s", flush=True)
a_, b_ = res["fp32 MFMA"], res["split-bf16 x6"]
print("bwd x6 vs fp32: dZ", [f"{rel(b_[0][l], a_[0][l].to(d)):.1e}" for l in range(4)], f"dE {rel(b_[1], a_[1].to(d)):.1e} dscale {rel(b_[2], a_[2].to(d)):.1e}")
