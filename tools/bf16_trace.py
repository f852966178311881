"""kernel-trace target: the headline batch (16 CylinderFlow meshes, 15 rounds, latent 128) in the bf16 matrix mode"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import graph_physics_amd as gp
from graph_physics_amd import harness, ops
dev = torch.device("cuda:0")
cfg = gp.cylinder_config(15, 128)
cfg["training"]["enable_vram_optimizations"] = True
eng = harness.Engine(cfg, dev, learning_rate=1e-4, num_steps=10000, warmup=100)
b = gp.cylinder_batch(16, 1885, 0).to(dev)
b.mgn_topology = ops.Topology(b.edge_index, int(b.x.shape[0]))
for _ in range(8):
    eng.train_step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.train_step(b)
torch.cuda.synchronize()
print(f"bf16 cylinder batch16: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per training step", flush=True)
