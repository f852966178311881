"""kernel-trace target: a few configs[4] training steps in one matrix mode (argv[1] = fp32 | bf16)"""
import os
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scipy.spatial import Delaunay
import graph_physics_amd as gp
from graph_physics_amd import harness, ops, preprocess as PP, transformer as T
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda:0")
n = 150000
pts = np.random.default_rng(0).random((n, 3)).astype(np.float32)
ei = PP.faces_to_edges(torch.from_numpy(Delaunay(pts).simplices.T.astype(np.int64)).to(dev), n)
cfg = {"model": {"type": "transformer", "message_passing_num": 10, "hidden_size": 64, "node_input_size": 14, "output_size": 3, "edge_input_size": 0, "num_heads": 4}, "training": {"use_temporal_block": False}}
torch.manual_seed(0)
net = gp.get_model(cfg).to(dev)
g = gp.Graph(x=torch.randn(n, 23, device=dev), edge_index=ei, pos=torch.from_numpy(pts).to(dev))
if os.environ.get("C5_PIN_TOPOLOGY"):   # the caller's numbering (no Morton renumbering inside the engine)
    g.mgn_attn_topology = T.get_attn_topology(ei, n)
tgt = torch.randn(n, 3, device=dev)
opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)
ops.set_matrix_precision(mode)
for _ in range(6):
    loss = ((net(g) - tgt) ** 2).mean(); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
