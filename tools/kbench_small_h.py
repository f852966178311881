#!/usr/bin/env python3
"""The edge launch of the reference's shipped config (training_config/cylinder.json: latent 32, one 1885-node mesh) on the generic
kernels, timed back to back: is the 60 us seen in the step (profiles/r05_shipped_kernel_stats.csv) the kernel or the idle GPU?
python tools/kbench_small_h.py [H] [batch]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, graph_physics_amd as gp
from graph_physics_amd import ops
dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = gp.cylinder_batch(nb, 1885, 0).to(dev)
topo = ops.Topology(g.edge_index, g.x.shape[0])
N, E = topo.N, topo.E
f = dict(dtype=torch.float32, device=dev)
x, e = torch.randn(N, H, **f), torch.randn(E, H, **f)
W0 = torch.randn(H, 3 * H, **f) * 0.1
Wh = [torch.randn(H, H, **f) * 0.15 for _ in range(3)]
bs = [torch.zeros(H, **f) for _ in range(4)]
sc = torch.ones(H, **f)


def mk():
    return dict(e_new=torch.empty(E, H, **f), m=torch.empty(E, H, **f), He=[torch.empty(E, H, **f) for _ in range(3)],
                Ue=torch.empty(E, H, **f), Re=torch.empty(E, **f))


sets = [mk() for _ in range(4)]


def launch(s, save):
    ops.mlp_fwd(E, H, [(e, None, H), (x, topo.dst_s, H), (x, topo.src_s, H)], [W0] + Wh, bs, sc, H, e, s["e_new"], s["m"],
                s["He"] if save else None, s["Ue"] if save else None, s["Re"] if save else None)


def node(s, save):
    ops.mlp_fwd(N, H, [(x, None, H), (x, None, H)], [W0[:, :2 * H].contiguous()] + Wh, bs, sc, H, x, s["e_new"][:N], None,
                [t[:N] for t in s["He"]] if save else None, s["Ue"][:N] if save else None, s["Re"][:N] if save else None)


for name, fn in (("edge fwd", launch), ("node fwd", node)):
    for save in (False, True):
        for i in range(20):
            fn(sets[i % 4], save)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(200):
                fn(sets[i % 4], save)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 5)
        print("%s H=%d E=%d N=%d save=%d: %.1f us per launch back to back (min of 5 x 200), median %.1f" % (name, H, E, N, save, min(ts), sorted(ts)[2]),
              flush=True)
