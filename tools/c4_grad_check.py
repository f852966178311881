#!/usr/bin/env python3
"""Per-parameter gradient error of the large-mesh step against the closure oracle (tests/test_hip_round3.py's
construction), for several mesh sizes / activations / recompute modes: finds where a size-dependent defect starts.
usage: python tools/c4_grad_check.py N[,N...] [relu|silu|both]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import recipe as R
import graph_physics_amd as gp
from graph_physics_amd import layers, ops
from oracle import mgn_oracle as O
from test_hip_round3 import _closure

dev = torch.device("cuda:0")
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1000000").split(",")]
acts = {"both": ["relu", "silu"]}.get(sys.argv[2] if len(sys.argv) > 2 else "both", [sys.argv[2] if len(sys.argv) > 2 else "relu"])
L = 2
for N in sizes:
    g = gp.square_mesh(N, seed=0)
    ei = g.edge_index
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 6)
    x_in = torch.randn(N, 11, generator=torch.Generator().manual_seed(2))
    seeds = np.concatenate([np.arange(0, 300), np.arange(N // 2, N // 2 + 300), np.arange(N - 300, N)])
    cot = R.randn((seeds.size, 2), 8)
    nodes, kept, loc, sub_ei = _closure(ei, N, seeds, L)
    for act in acts:
        P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        ref = O.epd_forward(x_in[nodes], g.edge_attr[torch.from_numpy(kept)], sub_ei, P, L, act=act)
        (ref[loc[seeds]] * cot).sum().backward()
        layers.set_use_silu_activation(act == "silu")
        net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
        layers.set_use_silu_activation(False)
        net.load_state_dict(params)
        for ren in ("off", "on"):
            ops.set_node_renumbering(ren)
            for rec in ("off", "on"):
                ops.set_activation_recompute(rec)
                graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=ei.to(dev), pos=g.pos.to(dev))
                net.zero_grad(set_to_none=True)
                out = net(graph)
                (out[torch.from_numpy(seeds).to(dev)] * cot.to(dev)).sum().backward()
                errs = {k: float((p.grad.cpu() - P[k].grad).abs().max() / P[k].grad.abs().max()) for k, p in net.named_parameters()}
                bad = {k: v for k, v in errs.items() if not v < 1e-3}
                fe = float((out.detach().cpu()[seeds] - ref.detach()[loc[seeds]]).abs().max() / ref.detach().abs().max())
                print(f"N={N} E={ei.shape[1]} act={act} renumber={ren} recompute={rec}: fwd {fe:.1e} worst grad {max(errs.values()):.2e} "
                      f"bad {len(bad)}/{len(errs)} {list(bad.items())[:6]}", flush=True)
                del graph, out
                torch.cuda.empty_cache()
        ops.set_activation_recompute("auto")
        ops.set_node_renumbering("auto")
