#!/usr/bin/env python3
"""Where the remaining torch (non-engine) launches of a training step come from: a TorchDispatchMode that logs every aten op
that launches a kernel (copy_, fill_, pad, clone, index, ...) with the innermost repo frame.  usage: step_ops_where.py [batch]"""
import os, sys, traceback, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, graph_physics_amd as gp
from graph_physics_amd import harness, ops
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda:0")
eng = harness.Engine(gp.cylinder_config(15, 128), dev, learning_rate=1e-4, num_steps=10000, warmup=100)
batch = gp.cylinder_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 1, 1885, 0).to(dev)
batch.mgn_topology = ops.Topology(batch.edge_index, batch.x.shape[0])
for _ in range(3): eng.train_step(batch)
torch.cuda.synchronize()
seen = collections.Counter()
SKIP = ("aten.view", "aten.detach", "aten.t.", "aten.slice", "aten.select", "aten._unsafe_view", "aten.expand", "aten.as_strided", "aten.alias",
        "aten.unsqueeze", "aten.squeeze", "aten.transpose", "aten.permute", "aten.reshape", "aten.empty", "aten._local_scalar", "aten.is_", "aten.sym_",
        "aten.stride", "aten.size", "aten.numel", "aten.new_empty", "aten.lift_fresh", "aten.unbind", "aten.split")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack() if "/graph-physics_amd/" in f.filename or "/graph_physics_amd/" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), None)
            seen[(name, where, shp)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    eng.train_step(batch)
torch.cuda.synchronize()
for (name, where, shp), n in sorted(seen.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{n:3d}  {name:34s} {where:28s} {shp}")
