"""training_config/cylinder.json as shipped (5 rounds, latent 32) on the 16-mesh batch: eager against hipGraph replay, training step and rollout step"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import graph_physics_amd as gp
from graph_physics_amd import harness, ops
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Lr = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = harness.Engine(gp.cylinder_config(Lr, H), dev, learning_rate=1e-4, num_steps=10000, warmup=100)
b = gp.cylinder_batch(16, 1885, 0).to(dev)
b.mgn_topology = ops.Topology(b.edge_index, int(b.x.shape[0]))
def t(fn, k):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
for _ in range(8): eng.train_step(b)
print(f"H={H} L={Lr} eager train  {t(lambda: eng.train_step(b), 30):.3f} ms", flush=True)
frames = [b] * 20
eng.rollout(frames[:3])
print(f"H={H} L={Lr} eager rollout {t(lambda: eng.rollout(frames), 3) / 20:.3f} ms per step", flush=True)
eng.capture_train_step(b, warmup=3)
print(f"H={H} L={Lr} graph train  {t(lambda: eng.train_step_graphed(None), 100):.3f} ms", flush=True)
eng.capture_rollout_step(b)
eng.rollout_graphed(frames[:3])
print(f"H={H} L={Lr} graph rollout {t(lambda: eng.rollout_graphed(frames), 3) / 20:.3f} ms per step", flush=True)
