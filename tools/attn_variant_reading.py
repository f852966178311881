#!/usr/bin/env python3
"""The three parity readings of the 10-block coarse-aneurysm forward (tests/test_transformer.py, N = 12000) for the build named by MGN_LIB."""
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R0, os.path.join(R0, "tests"), os.path.join(R0, "tests", "golden")]
import torch
import recipe as R
from conftest import rel_err, rms_err, elem_err
from oracle import mgn_oracle as O
import graph_physics_amd as gp
import test_transformer as TT
dev = torch.device("cuda:0")
for N, seed in ((12000, 606), (12000, 607), (20000, 608)):
    pos, ei, _ = R.delaunay_graph(N, seed, dim=3)
    net = gp.get_model(TT.ANEURYSM).to(dev)
    params = R.variant_params(net.state_dict(), seed)
    net.load_state_dict(params)
    x_in = R.randn((N, 23), seed + 1)
    with torch.no_grad():
        ref = O.etd_forward(x_in, ei, params, 10, 4)
        out = net(gp.Graph(x=x_in.to(dev), edge_index=ei.to(dev), pos=pos.to(dev)))
    print(f"{os.environ.get('MGN_LIB', 'shipped'):>28s} N={N} seed={seed}: max-rel {rel_err(out, ref):.2e} rms {rms_err(out, ref):.2e} element-wise {elem_err(out, ref):.2e}", flush=True)
