#!/usr/bin/env python3
"""The use_gated_mlp GraphNetBlock variant (layers.py:213-278,932-942) at the bench batch (16 x 1885 nodes): forward and
forward + backward of a 15-round EncodeProcessDecode with gated blocks next to the default blocks -- HIP-event time per step.
No shipped JSON uses the variant; this is its only performance record.  usage: python tools/kbench_gated_block.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_physics_amd as gp
from tools.kbench import timeit
dev = torch.device("cuda:0")
g = gp.cylinder_batch(16, 1885, 0).to(dev)
N, E = g.x.shape[0], g.edge_index.shape[1]
x_in, e_in = torch.randn(N, 11, device=dev), g.edge_attr
for gated in (False, True):
    torch.manual_seed(0)
    net = gp.EncodeProcessDecode(15, 11, 3, 2, hidden_size=128, use_gated_mlp=gated).to(dev)
    graph = gp.Graph(x=x_in, edge_attr=e_in, edge_index=g.edge_index, pos=g.pos)
    def fwd():
        with torch.no_grad():
            net(graph)
    def train():
        net.zero_grad(set_to_none=True)
        net(graph).square().mean().backward()
    tf, tt = timeit(fwd, iters=5, warm=2, rounds=3), timeit(train, iters=5, warm=2, rounds=3)
    nparam = sum(p.numel() for p in net.parameters())
    print(f"{'gated-MLP blocks' if gated else 'default blocks   '}: N={N} E={E} params {nparam / 1e6:.2f} M  forward {tf:7.2f} ms  forward + backward {tt:7.2f} ms", flush=True)
