#!/usr/bin/env python3
"""Which torch (non-engine) ops still run inside one training step: torch.profiler table of
CPU-side op names with their device time, engine kernels excluded."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, graph_physics_amd as gp
from graph_physics_amd import harness, ops
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
eng = harness.Engine(gp.cylinder_config(15, 128), dev, learning_rate=1e-4, num_steps=10000, warmup=100)
batch = gp.cylinder_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1885, 0).to(dev)
batch.mgn_topology = ops.Topology(batch.edge_index, batch.x.shape[0])
for _ in range(3): eng.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    eng.train_step(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0 and not e.key.startswith("k_") and "mgn" not in e.key]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.count:4d} {e.device_time_total:9.1f} us  {e.key[:90]}")
