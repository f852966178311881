#!/usr/bin/env python3
"""Where do non-zero gradient rows appear when the cotangent lives on a few seed nodes?  (debug of tools/c4_grad_check.py)"""
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
from test_hip_round3 import _closure

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
act = sys.argv[2] if len(sys.argv) > 2 else "silu"
L = 2
ops.set_node_renumbering("off")
g = gp.square_mesh(N, seed=0)
ei = g.edge_index
params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 6)
x_in = torch.randn(N, 11, generator=torch.Generator().manual_seed(2))
seeds = np.concatenate([np.arange(0, 300), np.arange(N // 2, N // 2 + 300), np.arange(N - 300, N)])
cot = R.randn((seeds.size, 2), 8)
nodes, kept, loc, sub_ei = _closure(ei, N, seeds, L)
in_closure = torch.zeros(N, dtype=torch.bool)
in_closure[torch.from_numpy(nodes)] = True
edge_in = torch.zeros(ei.shape[1], dtype=torch.bool)
edge_in[torch.from_numpy(kept)] = True
layers.set_use_silu_activation(act == "silu")
net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
layers.set_use_silu_activation(False)
net.load_state_dict(params)
graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=ei.to(dev))
topo = ops.Topology(graph.edge_index, N)
graph.mgn_topology = topo
caught = {}


def hook(name, module):
    def fn(mod, inp, out):
        out.register_hook(lambda gr: caught.__setitem__(name, gr.detach().clone()))
        return None
    module.register_forward_hook(fn)


hook("nodes_encoder.out", net.nodes_encoder)
hook("edges_encoder.out", net.edges_encoder)
def _pre(mod, inp):
    inp[0].register_hook(lambda gr: caught.__setitem__("decoder.in", gr.detach().clone()))


net.decode_module.register_forward_pre_hook(_pre)
out = net(graph)
(out[torch.from_numpy(seeds).to(dev)] * cot.to(dev)).sum().backward()
torch.cuda.synchronize()
for name, gr in caught.items():
    gr = gr.cpu()
    rows = gr.abs().amax(dim=1)
    if gr.shape[0] == N:
        outside = rows[~in_closure]
    else:  # edge rows are in dst-sorted order
        outside = rows[~edge_in[topo.perm_dst.long().cpu()]]
    print(f"{name}: shape {tuple(gr.shape)} max {float(rows.max()):.3e}; rows that must be zero: {int((outside != 0).sum())} non-zero of {outside.numel()}, "
          f"max {float(outside.max()) if outside.numel() else 0:.3e}, nan {int(torch.isnan(gr).sum())}")
