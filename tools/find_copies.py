#!/usr/bin/env python3
"""Where do the device copies and fills of one training step come from?  torch.profiler with stacks over one
eager step of the bench configuration; prints the python call sites of aten::copy_ / fill_ / zero_ (CUDA side)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, graph_physics_amd as gp
from graph_physics_amd import harness, ops
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
eng = harness.Engine(gp.cylinder_config(15, 128), dev, learning_rate=1e-4, num_steps=10000, warmup=100)
batch = gp.cylinder_batch(16, 1885, 0).to(dev)
batch.mgn_topology = ops.Topology(batch.edge_index, batch.x.shape[0])
for _ in range(3): eng.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.train_step(batch)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::contiguous", "aten::empty_like"):
        st = [s for s in (ev.stack or []) if "graph-physics_amd" in s or "graph_physics_amd" in s or "bench.py" in s]
        sites[(ev.name, st[0] if st else "(no package frame)")] += 1
for (name, site), n in sites.most_common(40):
    print("%4d  %-18s %s" % (n, name, site))
