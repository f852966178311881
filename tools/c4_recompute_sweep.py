#!/usr/bin/env python3
"""The 1M-node training step on one GPU against the number of rounds whose activations are recomputed in the backward pass
(ops.set_activation_recompute(k): the first k rounds keep only their inputs).  usage: python tools/c4_recompute_sweep.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, graph_physics_amd as gp
from graph_physics_amd import harness, ops, mesh
dev = torch.device("cuda:0")
g = mesh.square_mesh(1_000_000, 0)
n = g.x.shape[0]
torch.manual_seed(0)
net = gp.EncodeProcessDecode(15, 11, 3, 2, hidden_size=128).to(dev)
x_in = torch.randn(n, 11, generator=torch.Generator().manual_seed(1))
graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=g.edge_index.to(dev), pos=g.pos.to(dev))
tg, nt = torch.randn(n, 2, generator=torch.Generator().manual_seed(2)).to(dev), torch.zeros(n, device=dev)
opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)
def step():
    loss = harness.l2_loss(net(graph), tg, nt)
    opt.zero_grad(); loss.backward(); opt.step()
for k in ([int(a) if a.isdigit() else a for a in sys.argv[1:]] or [15, 11, 7, "auto"]):
    ops.set_activation_recompute(k)
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats(dev)
    step(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize()
    print(f"recompute {k}: {(time.perf_counter() - t) / 3 * 1e3:7.1f} ms/step  peak {torch.cuda.max_memory_allocated(dev) / 2**30:.0f} GiB", flush=True)
