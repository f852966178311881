#!/usr/bin/env python3
"""k_wgrad_x6 on the bench shape: the four E-row weight-gradient jobs of one round as ONE launch (+ k_wgrad_red), HIP-event timing.
With MGN_LIB naming a timing-experiment build (-DWGX_EXP_NOSPLIT / -DWGX_EXP_NOMFMA: results garbage by design) it prices the parts.
usage: python tools/kbench_wgrad.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_physics_amd as gp
from graph_physics_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = gp.cylinder_batch(nb, 1885, 0).to(dev)
E, H = g.edge_index.shape[1], 128
f = dict(dtype=torch.float32, device=dev)
torch.manual_seed(0)
dZ = [torch.randn(E, H, **f) for _ in range(4)]
X = [torch.randn(E, H, **f) for _ in range(4)]
gW = [torch.empty(H, 3 * H, **f)] + [torch.empty(H, H, **f) for _ in range(3)]
gb = [torch.empty(H, **f) for _ in range(4)]
nbk = H // 16
def fn():
    ops.wgrad([(dZ[0], H, nbk, X[0], H, nbk, H, gW[0], 0, 3 * H, gb[0])] + [(dZ[l], H, nbk, X[l], H, nbk, H, gW[l], 0, H, gb[l]) for l in range(1, 4)], dev)
t = timeit(fn)
byt = 8 * E * H * 4
print(f"{os.environ.get('MGN_LIB', 'shipped build'):>32s}: E={E}  {t * 1e3:7.1f} us  ({byt / t / 1e6:.0f} GB/s of operand rows)")
