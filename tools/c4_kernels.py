#!/usr/bin/env python3
"""The kernels of BASELINE configs[3] (1M-node / 6M-edge Delaunay mesh) on one GPU, as a command for rocprofv3 (VERDICT r3 item 4:
the north-star's scatter-add roofline needs profiler evidence at the size that leaves the 256 MiB Infinity Cache):
  what = "seg":  k_segsum<8> (forward scatter-add, 3.6 GB per launch) and k_segsum2<8> (both backward scatters) on their own
  what = "fwd":  whole-mesh inference forward passes (edge / node kernels at 1M nodes)
  what = "all":  both
After a 1 GiB device copy (the FETCH_SIZE calibration the PMC summaries use).
usage: python tools/c4_kernels.py [seg|fwd|all] [nodes=1000000]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import graph_physics_amd as gp
from graph_physics_amd import ops

what = sys.argv[1] if len(sys.argv) > 1 else "all"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dev = torch.device("cuda:0")
g = gp.square_mesh(n, seed=0)
E, H = int(g.edge_index.shape[1]), 128
a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
b.copy_(a)   # calibration: 1 GiB read + 1 GiB written
del a, b
torch.cuda.synchronize()
ei = g.edge_index.to(dev)
pos = g.pos.to(dev)
topo = ops.Topology(ei, n, renumber="morton", pos=pos)
print(f"N={n} E={E}", flush=True)
if what in ("seg", "all"):
    m = torch.randn(E, H, device=dev)
    agg, agg2 = torch.empty(n, H, device=dev), torch.empty(n, H, device=dev)
    for _ in range(4):
        ops.segsum(m, topo.rowptr_dst, None, agg)
    for _ in range(4):
        ops.segsum2(m, topo.rowptr_dst, None, agg, topo.rowptr_src, topo.perm_src, agg2)
    torch.cuda.synchronize()
    del m, agg, agg2
    torch.cuda.empty_cache()
if what in ("fwd", "all"):
    torch.manual_seed(0)
    net = gp.EncodeProcessDecode(15, 11, 3, 2, hidden_size=H).to(dev)
    graph = gp.Graph(x=torch.randn(n, 11, device=dev), edge_attr=g.edge_attr.to(dev), edge_index=ei, pos=pos)
    graph.mgn_topology = topo
    with torch.no_grad():
        for _ in range(2):
            net(graph)
    torch.cuda.synchronize()
print("done", flush=True)
