#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes: a calibration copy of known size, the segment-sum
and the training-mode edge kernel, each launched a few times (run under rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, graph_physics_amd as gp
from graph_physics_amd import ops
dev = torch.device("cuda:0")
g = gp.cylinder_batch(16, 1885, 0).to(dev)
topo = ops.Topology(g.edge_index, g.x.shape[0])
N, E, H = topo.N, topo.E, 128
f = dict(dtype=torch.float32, device=dev)
torch.manual_seed(0)
x, e = torch.randn(N, H, **f), torch.randn(E, H, **f)
Pd, Ps = torch.randn(N, H, **f), torch.randn(N, H, **f)
W0 = torch.randn(H, 3 * H, **f) * 0.05
Wh = [torch.randn(H, H, **f) * 0.09 for _ in range(3)]
bs = [torch.zeros(H, **f) for _ in range(4)]
sc = torch.ones(H, **f)
m, e_new, agg = torch.empty(E, H, **f), torch.empty(E, H, **f), torch.empty(N, H, **f)
He = [torch.empty(E, H, **f) for _ in range(3)]
Ue, Re = torch.empty(E, H, **f), torch.empty(E, **f)
big = torch.randn(1 << 28, **f)      # 1 GiB: past the 256 MiB Infinity Cache
dst = torch.empty_like(big)
for _ in range(3):
    dst.copy_(big)                    # calibration: reads 1 GiB, writes 1 GiB
for _ in range(5):
    ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, m, He, Ue, Re, ldw0=3 * H,
                adds=[(Pd, topo.dst_s), (Ps, topo.src_s)])
    ops.segsum(m, topo.rowptr_dst, None, agg)
torch.cuda.synchronize()
print("N", N, "E", E)
