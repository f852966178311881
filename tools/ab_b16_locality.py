#!/usr/bin/env python3
"""A/B at the bench configuration (16 x 1885 nodes): the meshes as generated (uniform points: ids carry no locality) against the same
meshes with every mesh's nodes numbered along its own Morton curve (done on the host here: an experiment on the input, not the engine).
Question: do the gathered Pd / Ps rows of the edge kernels cost time at this size, where every row is L2/MALL-resident?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, graph_physics_amd as gp
from graph_physics_amd import harness, ops, mesh, partition
dev = torch.device("cuda:0")

def local_mesh(n, seed):
    g = mesh.cylinder_mesh(n, seed)
    order = np.argsort(partition.morton_keys(g.pos.numpy()), kind="stable")
    rank = np.empty_like(order); rank[order] = np.arange(n)
    o = torch.from_numpy(order)
    ei = torch.from_numpy(rank)[g.edge_index]
    return gp.Graph(x=g.x[o], y=g.y[o], pos=g.pos[o], face=torch.from_numpy(rank)[g.face], edge_index=ei, edge_attr=mesh.edge_features(g.pos[o], ei))

def run(batch, tag):
    torch.manual_seed(0)
    eng = harness.Engine(gp.cylinder_config(15, 128), dev, learning_rate=1e-4, num_steps=10000, warmup=100)
    batch = batch.to(dev)
    batch.mgn_topology = ops.Topology(batch.edge_index, batch.x.shape[0])
    for _ in range(8):
        eng.train_step(batch)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30):
        eng.train_step(batch)
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t) / 30 * 1e3:.3f} ms/step", flush=True)

raw = gp.cylinder_batch(16, 1885, 0)
loc = mesh.collate([local_mesh(1885, i) for i in range(16)])
if len(sys.argv) > 1:   # PMC passes: one variant only
    run(raw if sys.argv[1] == "raw" else loc, sys.argv[1])
else:
    for _ in range(2):
        run(raw, "generator numbering"), run(loc, "per-mesh Morton numbering")
