"""kernel-trace target: rollout steps of training_config/cylinder.json as shipped (5 rounds, latent 32), 16-mesh batch, hipGraph replay"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import graph_physics_amd as gp
from graph_physics_amd import harness, ops
dev = torch.device("cuda:0")
eng = harness.Engine(gp.cylinder_config(5, 32), dev, learning_rate=1e-4, num_steps=10000, warmup=100)
b = gp.cylinder_batch(16, 1885, 0).to(dev)
b.mgn_topology = ops.Topology(b.edge_index, int(b.x.shape[0]))
eng.capture_rollout_step(b)
frames = [b] * 40
eng.rollout_graphed(frames[:3])
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.rollout_graphed(frames)
torch.cuda.synchronize()
print(f"shipped rollout (graph): {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per step", flush=True)
