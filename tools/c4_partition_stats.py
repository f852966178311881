#!/usr/bin/env python3
"""Partition statistics of the BASELINE configs[3] mesh (1M-node Delaunay, 8-way RCB): rows
owned, ghost rows and bytes exchanged per round -- host-only, no GPU needed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_physics_amd as gp
from graph_physics_amd import partition as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time(); g = gp.square_mesh(n, seed=0); print(f"mesh: N={n} E={g.edge_index.shape[1]} ({time.time()-t0:.1f}s)")
t0 = time.time(); part = P.rcb_partition(g.pos.numpy(), k); print(f"rcb {k}-way: {time.time()-t0:.1f}s  edge cut {100*P.edge_cut(g.edge_index, part):.2f}%")
for r in range(k):
    pl = P.build_rank_plan(g.edge_index, part, r, k)
    peers = sum(1 for c in pl.send_counts if c > 0)
    print(f"rank {r}: owned {pl.n_own}  edges {pl.edge_ids.numel()}  ghosts {pl.n_ghost} ({100*pl.n_ghost/pl.n_own:.2f}%)  "
          f"send {pl.send_idx.numel()} rows = {pl.send_idx.numel()*512/1e6:.2f} MB/round to {peers} peers")
