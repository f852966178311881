#!/usr/bin/env python3
"""BASELINE configs[3] end to end on N ranks: a synthetic Delaunay mesh (default 1M nodes / 6M
directed edges), recursive-coordinate-bisection node partition, one-hop halo exchange of ghost
latents per round, training steps (forward + global masked L2 + backward + summed gradient
all-reduce + fused clip/AdamW) and inference steps.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29533 tools/c4_multirank.py [--nodes 1000000] [--steps 5]

One process per GPU over RCCL.  Rehearsal on a 1-GPU box: MGN_DIST_BACKEND=gloo MGN_SHARE_GPU=1
(the ranks share the device and the collectives go through the host: correctness of the path, not
its speed).  Prints one JSON line on rank 0: node*train-steps/s and node*steps/s of the whole mesh."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import graph_physics_amd as gp
from graph_physics_amd import distributed as D, harness, partition as P

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--rounds", type=int, default=15)
a = ap.parse_args()
rank, world, local = D.init_from_env()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
g = gp.square_mesh(a.nodes, seed=0)                       # every rank builds the same mesh (seeded)
N, E = a.nodes, g.edge_index.shape[1]
t0 = time.perf_counter()
part = P.rcb_partition(g.pos.numpy(), world)
plan = P.build_rank_plan(g.edge_index, part, rank, world)
t_part = time.perf_counter() - t0
torch.manual_seed(0)
net = gp.EncodeProcessDecode(a.rounds, 11, 3, 2, hidden_size=128).to(dev)
D.broadcast_parameters(net)
pm = D.PartitionedEPD(net, plan)
x_in = torch.randn(N, 11, generator=torch.Generator().manual_seed(1))
tgt = torch.randn(N, 2, generator=torch.Generator().manual_seed(2))
nt = torch.zeros(N)
xo, eo = x_in[plan.owned].to(dev), g.edge_attr[plan.edge_ids].to(dev)
to, no = tgt[plan.owned].to(dev), nt[plan.owned].to(dev)
opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)
sync = D.GradAllReduce(average=False)

def barrier():
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

def train_step():
    out = pm(xo, eo)
    loss = D.partitioned_loss(out, to, no)
    opt.zero_grad()
    loss.backward()
    sync(net.parameters())
    opt.step()
    return loss

def timed(fn, n):
    for _ in range(a.warmup):
        fn()
    barrier()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    barrier()
    dt = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return float(dt) / n

t_train = timed(train_step, a.steps)
with torch.no_grad():
    t_inf = timed(lambda: pm(xo, eo), a.steps)
lt = train_step().detach().clone()   # each rank holds its share of the global masked mean
if world > 1:
    dist.all_reduce(lt)
loss = float(lt)
if rank == 0:
    print(json.dumps({"workload": f"Delaunay mesh N={N} E={E}, {a.rounds} rounds, latent 128, {world}-way RCB partition + halo exchange",
                      "n_gpus": world, "backend": dist.get_backend() if world > 1 else "none",
                      "train_ms_per_step": round(1e3 * t_train, 2), "node_train_steps_per_s": round(N / t_train, 1),
                      "rollout_ms_per_step": round(1e3 * t_inf, 2), "node_steps_per_s": round(N / t_inf, 1),
                      "owned": plan.n_own, "ghosts": plan.n_ghost, "local_edges": int(plan.edge_ids.numel()),
                      "partition_s": round(t_part, 2), "loss": loss}), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
