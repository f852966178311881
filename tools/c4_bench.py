#!/usr/bin/env python3
"""BASELINE configs[3] on ONE GPU (the 8-GPU run is the driver's): (a) rollout of the whole
1M-node / 6M-edge mesh, (b) forward+backward of ONE rank's share of the 8-way partition
(125k owned nodes, 750k edges; ghost rows zero-filled, no exchange) = the per-GPU compute of
the partitioned training step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_physics_amd as gp
from graph_physics_amd import ops, partition as P, distributed as D

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g = gp.square_mesh(n, seed=0)
E = g.edge_index.shape[1]
print(f"mesh N={n} E={E}", flush=True)
torch.manual_seed(0)
net = gp.EncodeProcessDecode(15, 11, 3, 2, hidden_size=128).to(dev)
x_in = torch.randn(n, 11)
e_in = g.edge_attr

def sync_time(fn, iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

# (a) whole-mesh inference on one GPU
graph = gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=g.edge_index.to(dev))
graph.mgn_topology = ops.Topology(graph.edge_index, n)
with torch.no_grad():
    t = sync_time(lambda: net(graph), 3)
print(f"(a) 1-GPU forward of the whole mesh: {t*1e3:.1f} ms -> {n/t/1e6:.2f} M node-steps/s  (peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB)", flush=True)
del graph; torch.cuda.empty_cache()

# (b) one rank's share of the 8-way partition: forward + backward (compute only)
part = P.rcb_partition(g.pos.numpy(), 8)
plan = P.build_rank_plan(g.edge_index, part, 0, 8)
plan.world = 1  # no process group here: ghost rows are zero-filled (timing only)
pm = D.PartitionedEPD(net, plan)
xo, eo = x_in[plan.owned].to(dev), e_in[plan.edge_ids].to(dev)
def step():
    out = pm(xo, eo)
    out.square().mean().backward()
torch.cuda.reset_peak_memory_stats()
t = sync_time(step, 3)
print(f"(b) rank 0 of 8: owned {plan.n_own}, ghosts {plan.n_ghost}, edges {plan.edge_ids.numel()}: fwd+bwd {t*1e3:.1f} ms "
      f"(peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB) -> 8-GPU estimate {n/t/1e6:.2f} M node-train-steps/s before comms", flush=True)
