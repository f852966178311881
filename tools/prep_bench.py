#!/usr/bin/env python3
"""Measurement of the N1 / N2 rows: on-device edge construction for the C4 mesh
(1M nodes / 2M triangles) and the fused Simulator pre / post processing on the bench batch,
with the CPU statement of the same work (the oracle; numpy / torch on the host) beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import graph_physics_amd as gp
from graph_physics_amd import preprocess as P
from oracle import mgn_oracle as O

dev = torch.device("cuda:0")

def gpu_ms(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
from scipy.spatial import Delaunay
pts = np.random.default_rng(0).random((n, 2)).astype(np.float32)
face = Delaunay(pts).simplices.T.astype(np.int64)
F = face.shape[1]
fd, pd = torch.from_numpy(face).to(dev), torch.from_numpy(pts).to(dev)
t0 = time.perf_counter(); ei_ref = O.faces_to_edges_oracle(face, n); t_cpu = time.perf_counter() - t0
ei = P.faces_to_edges(fd, n)
assert np.array_equal(ei.cpu().numpy(), ei_ref)
E = ei.shape[1]
t = gpu_ms(lambda: P.faces_to_edges(fd, n))
print(f"faces->edges  N={n} F={F} E={E}: {t:.2f} ms on the GPU (incl. the one sync that reads E), {t_cpu*1e3:.0f} ms numpy oracle on the host")
t0 = time.perf_counter(); ea_ref = O.edge_features_oracle(torch.from_numpy(pts), torch.from_numpy(ei_ref)); t_cpu = time.perf_counter() - t0
t = gpu_ms(lambda: P.edge_features(pd, ei), 20)
byts = E * (16 + 12) + E * 16  # idx pair + out row + two gathered positions
print(f"edge features E={E}: {t*1e3:.1f} us = {byts/t/1e6:.0f} GB/s ({byts/t/1e6/8000*100:.0f}% of 8 TB/s), {t_cpu*1e3:.0f} ms torch oracle on the host")

cfg = gp.cylinder_config(15, 128)
batch = gp.cylinder_batch(16, 1885, 0).to(dev)
sim = gp.get_simulator(cfg, gp.get_model(cfg).to(dev), dev)
sim.train()
for fused in (True, False):
    sim.fused = fused
    t = gpu_ms(lambda: sim._build_input_graph(batch, True), 20)
    Nn, Ee = batch.x.shape[0], batch.edge_attr.shape[0]
    byts = 2 * (Nn * 4 * 4 + Nn * 2 * 4 + Ee * 3 * 4) + (Nn * 11 + Nn * 2 + Ee * 3) * 4  # two reads (statistics, normalise) + one write
    print(f"simulator pre (training, b16)  {'fused HIP' if fused else 'torch ops':9s}: {t*1e3:7.1f} us" + (f" = {byts/t/1e6:.0f} GB/s" if fused else ""))
sim.eval()
no = torch.randn(batch.x.shape[0], 2, device=dev)
with torch.no_grad():
    for fused in (True, False):
        sim.fused = fused
        t = gpu_ms(lambda: sim.predict(batch, no, mask_truth=True), 20)
        print(f"simulator post + re-imposition {'fused HIP' if fused else 'torch ops':9s}: {t*1e3:7.1f} us")
