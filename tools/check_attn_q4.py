#!/usr/bin/env python3
"""Four-head sparse attention: the quad-distributed form (Q4, default) against the every-lane form (MGN_ATTN_Q4=0) -- the same sums in another association -- and
their launch times on configs[4]'s mesh.  usage: python tools/check_attn_q4.py save|compare FILE [nodes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.spatial import Delaunay
from graph_physics_amd import preprocess as PP, transformer as T
from tools.kbench import timeit
dev = torch.device("cuda:0")
mode, path = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 150000
pts = np.random.default_rng(0).random((n, 3)).astype(np.float32)
ei = PP.faces_to_edges(torch.from_numpy(Delaunay(pts).simplices.T.astype(np.int64)).to(dev), n)
topo = T.get_attn_topology(ei, n, pos=torch.from_numpy(pts).to(dev), renumber=True)
out = {}
for H, nh in ((64, 4), (128, 4), (16, 4), (32, 4)):
    torch.manual_seed(H)
    for b16 in (False, True):
        q, k, v = (torch.randn(n, H, device=dev).requires_grad_(True) for _ in range(3))
        if b16:
            k, v = (t.detach().bfloat16().float().requires_grad_(True) for t in (k, v))
        dy = torch.randn(n, H, device=dev)
        y = T.sparse_attention(q, k, v, topo, nh, b16=b16)
        y.backward(dy)
        tag = f"H{H}_b16{int(b16)}"
        for nm, t in (("y", y), ("dq", q.grad), ("dk", k.grad), ("dv", v.grad)):
            out[f"{tag}_{nm}"] = t.detach().cpu().numpy()
        if H == 64:
            tf = timeit(lambda: T.sparse_attention(q.detach(), k.detach(), v.detach(), topo, nh, b16=b16))
            def fb():
                q.grad = k.grad = v.grad = None
                T.sparse_attention(q, k, v, topo, nh, b16=b16).backward(dy)
            tb = timeit(fb)
            print(f"MGN_ATTN_Q4={os.environ.get('MGN_ATTN_Q4', '1')} H=64 b16={int(b16)}: forward {tf * 1e3:.1f} us, forward + backward {tb * 1e3:.1f} us", flush=True)
if mode == "save":
    np.savez(path, **out)
else:
    ref = np.load(path)
    for b in (0, 1):   # (bf16 mode: a moved rounding decision of a bf16 result is 2^-9 of that element)
        worst = max(float(np.abs(out[k] - ref[k]).max() / np.abs(ref[k]).max()) for k in out if f"_b16{b}_" in k)
        print(f"b16={b}: largest difference to the other form, relative to the tensor's largest element: {worst:.2e}")
