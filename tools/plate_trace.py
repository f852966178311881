"""kernel-trace target: configs[2]-style training steps (16 plate meshes per step, bf16 matrix mode), as bench.py's plate_bf16.batch16"""
import sys
sys.path.insert(0, "/root/repo")
import torch
import graph_physics_amd as gp
from graph_physics_amd import harness, ops
dev = torch.device("cuda:0")
cfg = gp.plate_config(15, 128)
meshes = [gp.plate_mesh(1300, seed=61 + i, device=dev) for i in range(16)]
eng = harness.Engine(cfg, dev, learning_rate=1e-4, num_steps=10000, warmup=100)
b = gp.collate(meshes)
b.mgn_topology = ops.Topology(b.edge_index, int(b.x.shape[0]))
import time
for _ in range(8):
    eng.train_step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.train_step(b)
torch.cuda.synchronize()
print(f"plate bf16 batch16: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per training step", flush=True)
