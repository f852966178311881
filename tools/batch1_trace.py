"""kernel-trace target: one CylinderFlow mesh per training step under hipGraph replay (bench.py's batch1 record)"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import graph_physics_amd as gp
from graph_physics_amd import harness, ops
dev = torch.device("cuda:0")
eng = harness.Engine(gp.cylinder_config(15, 128), dev, learning_rate=1e-4, num_steps=10000, warmup=100)
b = gp.cylinder_batch(1, 1885, 0).to(dev)
b.mgn_topology = ops.Topology(b.edge_index, b.x.shape[0])
eng.capture_train_step(b, warmup=3)
for _ in range(10):
    eng.train_step_graphed(None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    eng.train_step_graphed(None)
torch.cuda.synchronize()
print(f"batch1: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per training step", flush=True)
