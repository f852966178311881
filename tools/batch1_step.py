#!/usr/bin/env python3
"""Workload for a kernel trace of the batch-1 training step (one mesh, eager launches, 5 steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, graph_physics_amd as gp
from graph_physics_amd import harness, ops
dev = torch.device("cuda:0")
eng = harness.Engine(gp.cylinder_config(15, 128), dev, learning_rate=1e-4, num_steps=10000, warmup=100)
b = gp.cylinder_batch(1, 1885, 0).to(dev)
b.mgn_topology = ops.Topology(b.edge_index, b.x.shape[0])
for _ in range(3): eng.train_step(b)
torch.cuda.synchronize()
print("MARK")
for _ in range(5): eng.train_step(b)
torch.cuda.synchronize()
