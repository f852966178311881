#!/usr/bin/env python3
"""k_linear_x6 (six-term split-bf16 dense launch) against an fp64 evaluation, next to the exact-fp32 MFMA path's error
(MGN_LINEAR_X6=0 in a second process).  usage: python tools/check_linear_x6.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from graph_physics_amd import dense as D, ops
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 70001
f = dict(dtype=torch.float32, device=dev)
torch.manual_seed(1)
d = torch.float64
def rel(a, b):
    return float((a.to(d) - b).abs().max() / b.abs().max())
def norm(x, sc):
    return sc.to(d) * x.to(d) / (x.to(d).norm(dim=1, keepdim=True) / x.shape[1] ** 0.5 + ops.EPS)
worst = 0.0
for K, N in ((64, 64), (64, 192), (192, 64), (128, 128), (32, 64), (384, 64), (256, 96)):
    x = torch.randn(M, K, **f) * torch.exp(torch.randn(M, 1, **f))
    W, b = torch.randn(N, K, **f) / K ** 0.5, torch.randn(N, **f)
    W2, b2 = torch.randn(N, K, **f) / K ** 0.5, torch.randn(N, **f)
    sc = torch.rand(K, **f) + 0.5
    res = torch.randn(M, N, **f)
    # plain + residual
    o = D.linear_launch(x, W, b, resid=res)
    e1 = rel(o, res.to(d) + x.to(d) @ W.to(d).t() + b.to(d))
    # norm prologue with side outputs, gelu
    inv, n_out, z1 = torch.empty(M, **f), torch.empty(M, K, **f), torch.empty(M, N, **f)
    o = D.linear_launch(x, W, b, norm_scale=sc, act=2, inv_out=inv, n_out=n_out, saveZ1=z1)
    n = norm(x, sc)
    z = n @ W.to(d).t() + b.to(d)
    e2 = max(rel(o, F.gelu(z)), rel(n_out, n), rel(z1, z))
    # gated product, silu, both saves
    z1, z2 = torch.empty(M, N, **f), torch.empty(M, N, **f)
    o = D.linear_launch(x, W, b, W2=W2, b2=b2, norm_scale=sc, act=1, saveZ1=z1, saveZ2=z2)
    zz2 = n @ W2.to(d).t() + b2.to(d)
    e3 = max(rel(o, F.silu(z) * zz2), rel(z2, zz2), rel(z1, z))
    # two phases, the second gathered
    e4 = 0.0
    if K % 32 == 0 and K >= 64:
        k1 = K // 2
        xa, xb = x[:, :k1].contiguous(), torch.randn(M // 3, K - k1, **f)
        idx = torch.randint(0, M // 3, (M,), dtype=torch.int32, device=dev)
        o = D.linear_launch(xa, W, None, x2=xb, idx=(None, idx, None), M=M)
        e4 = rel(o, xa.to(d) @ W[:, :k1].to(d).t() + xb.to(d)[idx.long()] @ W[:, k1:].to(d).t())
    # the input gradient of a Linear straight from its weight (staged transposed)
    dz = torch.randn(M, N, **f)
    o = D.input_gradient(dz, W, resid=x)
    e5 = rel(o, x.to(d) + dz.to(d) @ W.to(d))
    worst = max(worst, e1, e2, e3, e4, e5)
    print(f"{K:4d} -> {N:4d}: resid {e1:.2e}  norm+gelu+saves {e2:.2e}  gated silu {e3:.2e}  two phases / gather {e4:.2e}  dX = dZ W {e5:.2e}", flush=True)
print(f"MGN_LINEAR_X6={os.environ.get('MGN_LINEAR_X6', '1')}  rows {M}  worst {worst:.2e}")
