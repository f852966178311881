#!/usr/bin/env python3
"""Micro-benchmarks of the dense row launches (csrc/mgn_dense.hip) at the c5 record's size: HIP-event time and the rate of
their algorithmic bytes.  usage: python tools/kbench_dense.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from graph_physics_amd import _capi, ops
from graph_physics_amd import dense as D
from tools.kbench import timeit

dev = torch.device("cuda:0")
PREC = int(os.environ.get("KB_PREC", "0"))   # 1: the bf16 matrix mode of the Linear launches
LL = lambda *a_, **k_: D.linear_launch(*a_, precision=PREC, **k_)  # noqa: E731
M = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
f = dict(dtype=torch.float32, device=dev)
torch.manual_seed(0)
x64, x192 = torch.randn(M, 64, **f), torch.randn(M, 192, **f)
W = {(n, k): torch.randn(n, k, **f) * 0.1 for n, k in ((64, 64), (192, 64), (64, 192))}
b64, b192 = torch.randn(64, **f), torch.randn(192, **f)
sc = torch.rand(64, **f) + 0.5
o64, o192 = torch.empty(M, 64, **f), torch.empty(M, 192, **f)
inv, n_out = torch.empty(M, **f), torch.empty(M, 64, **f)
z1, z2 = torch.empty(M, 192, **f), torch.empty(M, 192, **f)


def rep(name, fn, nbytes):
    t = timeit(fn)
    print(f"{name:58s} {t * 1e3:8.1f} us   {nbytes / t / 1e6:7.0f} GB/s", flush=True)


row = 4.0 * M
rep("linear 64->64 plain", lambda: LL(x64, W[(64, 64)], b64, out=o64), row * 128)
rep("linear 64->64 norm prologue (inference)", lambda: LL(x64, W[(64, 64)], b64, norm_scale=sc, out=o64), row * 128)
rep("linear 64->64 norm prologue + inv + n_out (training)", lambda: LL(x64, W[(64, 64)], b64, norm_scale=sc, out=o64, inv_out=inv, n_out=n_out), row * 192)
rep("linear 64->64 + residual", lambda: LL(x64, W[(64, 64)], b64, resid=x64, out=o64), row * 192)
rep("gated 64->192 norm, gelu (inference)", lambda: LL(x64, W[(192, 64)], b192, W2=W[(192, 64)], b2=b192, norm_scale=sc, act=2, out=o192), row * 256)
rep("gated 64->192 norm, gelu + Z1 Z2 n_out (training)", lambda: LL(x64, W[(192, 64)], b192, W2=W[(192, 64)], b2=b192, norm_scale=sc, act=2, out=o192, inv_out=inv, n_out=n_out, saveZ1=z1, saveZ2=z2), row * (64 + 192 * 3 + 64))
rep("linear 192->64 + residual", lambda: LL(x192, W[(64, 192)], b64, resid=x64, out=o64), row * 320)
rep("linear 192->64 (dX of the gate Linears)", lambda: LL(x192, W[(64, 192)], None, out=o64), row * 256)
WT = torch.randn(192, 64, **f) * 0.1   # the nn.Linear weight of a 64 -> 192 Linear: its input gradient is 192 -> 64
rep("dX = dZ W, 192 -> 64, weight staged transposed", lambda: D.input_gradient(x192, WT, precision=PREC), row * 256)
rep("dX = dZ W, 192 -> 64, + residual", lambda: D.input_gradient(x192, WT, resid=x64, precision=PREC), row * 320)
L = _capi.lib()
dx, dscale = torch.empty(M, 64, **f), torch.empty(64, **f)
ws = torch.empty(L.mgn_rownorm_bwd_workspace_bytes(64), dtype=torch.uint8, device=dev)
arr = (_capi.RownormPhase * 1)()
arr[0].x, arr[0].ldx, arr[0].K, arr[0].idx, arr[0].dx, arr[0].lddx = x64.data_ptr(), 64, 64, None, dx.data_ptr(), 64
rep("rownorm_bwd 64 (+ dscale reduction)", lambda: L.mgn_rownorm_bwd(o64.data_ptr(), arr, 1, inv.data_ptr(), sc.data_ptr(), 1e-8, M, dscale.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream(dev)), row * 192)
y = torch.empty(M, 64, **f)
rep("rownorm_fwd 64", lambda: L.mgn_rownorm_fwd(x64.data_ptr(), 64, 64, sc.data_ptr(), 1e-8, M, y.data_ptr(), inv.data_ptr(), ops._stream(dev)), row * 128)
rep("act_gate_bwd 192 gelu", lambda: L.mgn_act_gate_bwd(o192.data_ptr(), z1.data_ptr(), z2.data_ptr(), M, 192, 2, 0, o192.data_ptr(), z2.data_ptr(), ops._stream(dev)), row * 192 * 5)
dW, db = torch.empty(192, 64, **f), torch.empty(192, **f)
for slab in (128, 64):
    jobs = []
    for j0 in range(0, 192, slab):
        nj = min(slab, 192 - j0)
        jobs.append((z1[:, j0:j0 + nj], 192, nj // 16, x64, 64, 4, 64, dW, j0 * 64, 64, db[j0:j0 + nj]))
    rep(f"wgrad dW[192,64] = dZ^T x, {slab}-row slabs ({len(jobs)} jobs)", lambda: ops.wgrad(jobs, dev), row * 256)
jobs = [(o64, 64, 4, x64, 64, 4, 64, torch.empty(64, 64, **f), 0, 64, b64)]
rep("wgrad dW[64,64]", lambda: ops.wgrad(jobs, dev), row * 128)
