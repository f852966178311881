#!/usr/bin/env python3
"""Workload for profiling BASELINE configs[4] (bench.py's c5 record): a few training steps and forwards of the coarse-aneurysm
Transformer on a 3-D mesh.  usage: [rocprofv3 --kernel-trace --stats -d OUT --] python3 tools/c5_step.py [nodes [sorted]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.spatial import Delaunay

import graph_physics_amd as gp
from graph_physics_amd import harness
from graph_physics_amd import preprocess as PP
from graph_physics_amd import transformer as T

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
pts = np.random.default_rng(0).random((n, 3)).astype(np.float32)
if len(sys.argv) > 2 and sys.argv[2] == "sorted":   # experiment on the input: points numbered along a Morton curve
    from graph_physics_amd import partition
    pts = pts[np.argsort(partition.morton_keys(pts), kind="stable")]
ei = PP.faces_to_edges(torch.from_numpy(Delaunay(pts).simplices.T.astype(np.int64)).to(dev), n)
cfg = {"model": {"type": "transformer", "message_passing_num": 10, "hidden_size": 64, "node_input_size": 14, "output_size": 3, "edge_input_size": 0,
                 "num_heads": 4}, "training": {"use_temporal_block": False}}
torch.manual_seed(0)
net = gp.get_model(cfg).to(dev)
g = gp.Graph(x=torch.randn(n, 23, device=dev), edge_index=ei, pos=torch.from_numpy(pts).to(dev))
if os.environ.get("C5_PIN_TOPOLOGY"):   # the caller's numbering (no Morton renumbering inside the engine)
    g.mgn_attn_topology = T.get_attn_topology(ei, n)
tgt = torch.randn(n, 3, device=dev)
opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)


def train():
    loss = ((net(g) - tgt) ** 2).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()


train()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    train()
torch.cuda.synchronize()
print(f"N={n} E={ei.shape[1]}: training step {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
with torch.no_grad():
    net(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        net(g)
    torch.cuda.synchronize()
print(f"forward {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
