#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` and, in a
second run, `--pmc WRITE_SIZE`): a 1 GiB calibration copy (past the 256 MiB Infinity Cache),
then two real training steps of the bench configuration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, graph_physics_amd as gp
from graph_physics_amd import harness, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
eng = harness.Engine(gp.cylinder_config(15, 128), dev, learning_rate=1e-4, num_steps=10000, warmup=100)
batch = gp.cylinder_batch(16, 1885, 0).to(dev)
batch.mgn_topology = ops.Topology(batch.edge_index, batch.x.shape[0])
big = torch.randn(1 << 28, dtype=torch.float32, device=dev)
dst = torch.empty_like(big)
for _ in range(3):
    dst.copy_(big)  # calibration: reads 1 GiB, writes 1 GiB
del big, dst
for _ in range(3):
    eng.train_step(batch)
torch.cuda.synchronize()
print("N", batch.x.shape[0], "E", batch.edge_index.shape[1])
