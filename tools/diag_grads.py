#!/usr/bin/env python3
"""Diagnostic: gradient errors of the HIP path vs the fp32 / fp64 oracle at depth, with the ReLU masks that
differ counted; the same net with SiLU (no masks) for comparison.  Prints one line per case."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import torch
import graph_physics_amd as gp
import recipe as R
from conftest import rel_err, rms_err
from oracle import mgn_oracle as O
import test_hip_configs as T

dev = torch.device("cuda:0")
from graph_physics_amd import _capi
import ctypes as C
occ = (C.c_int * 6)()
print("occupancy rc", _capi.lib().mgn_debug_occupancy(occ), list(occ), flush=True)
for name, g, L, seed in (("N=1885 L=15", gp.cylinder_mesh(1885, 0), 15, 77), ("N=1885 L=5", gp.cylinder_mesh(1885, 0), 5, 77),
                         ("N=400 L=15", gp.cylinder_mesh(400, 1), 15, 79)):
    out, grads, (o32, g32), (o64, g64), (flips, total, worst) = T._grad_case(dev, g, L, seed)
    ks = list(grads)
    e32 = max(rel_err(grads[k], g32[k]) for k in ks)
    h64 = max(rel_err(grads[k], g64[k]) for k in ks)
    c64 = max(rel_err(g32[k], g64[k]) for k in ks)
    r32 = max(rms_err(grads[k], g32[k]) for k in ks)
    rh64 = max(rms_err(grads[k], g64[k]) for k in ks)
    rc64 = max(rms_err(g32[k], g64[k]) for k in ks)
    print(f"{name}: fwd {rel_err(out, o32):.2e} | max-rel hip-vs-32 {e32:.2e} hip-vs-64 {h64:.2e} cpu32-vs-64 {c64:.2e} | rms hip-vs-32 {r32:.2e} "
          f"hip-vs-64 {rh64:.2e} cpu32-vs-64 {rc64:.2e} | flips {flips}/{total} worst|z|/max {worst:.1e}", flush=True)
# SiLU: smooth, no masks
gp.layers.set_use_silu_activation(True)
try:
    net = gp.EncodeProcessDecode(15, 11, 3, 2, hidden_size=128).to(dev)
finally:
    gp.layers.set_use_silu_activation(False)
g = gp.cylinder_mesh(1885, 0)
N, E = g.x.shape[0], g.edge_index.shape[1]
params = R.make_params(R.epd_param_shapes(15, 128, 11, 3, 2), 77)
net.load_state_dict(params)
x_in, e_in, cot = R.randn((N, 11), 78), R.randn((E, 3), 79), R.randn((N, 2), 80)
res = {}
for dt in (torch.float32, torch.float64):
    p = {k: v.clone().to(dt).requires_grad_(True) for k, v in params.items()}
    o = O.epd_forward(x_in.to(dt), e_in.to(dt), g.edge_index, p, 15, act="silu")
    (o * cot.to(dt)).sum().backward()
    res[dt] = (o.detach(), {k: v.grad for k, v in p.items()})
out = net(gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=g.edge_index.to(dev)))
(out * cot.to(dev)).sum().backward()
gr = {k: p.grad.cpu() for k, p in net.named_parameters()}
print(f"SiLU N=1885 L=15: fwd {rel_err(out, res[torch.float32][0]):.2e} | max-rel hip-vs-32 {max(rel_err(gr[k], res[torch.float32][1][k]) for k in gr):.2e} "
      f"hip-vs-64 {max(rel_err(gr[k], res[torch.float64][1][k]) for k in gr):.2e} cpu32-vs-64 "
      f"{max(rel_err(res[torch.float32][1][k], res[torch.float64][1][k]) for k in gr):.2e}", flush=True)
