#!/usr/bin/env python3
"""Do two of the step's kernels gain anything from running on two HIP streams?  The weight gradients of a round depend on the
backward chains of THAT round only, so they could run beside the next round's chain kernels.  Pairs timed back to back on one
stream and concurrently on two (K iterations each, total wall time per iteration):
  edge forward chain (training saves)  ||  weight gradients of a round (k_wgrad_pc)
  node-row chain (472 one-tile workgroups)  ||  weight gradients
usage: python tools/overlap_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_physics_amd as gp
from graph_physics_amd import ops, _capi
dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = gp.cylinder_batch(nb, 1885, 0).to(dev)
topo = ops.Topology(g.edge_index, g.x.shape[0])
N, E, H = topo.N, topo.E, 128
f = dict(dtype=torch.float32, device=dev)
torch.manual_seed(0)
x, e = torch.randn(N, H, **f), torch.randn(E, H, **f)
W0 = torch.randn(H, 3 * H, **f) * 0.05
Wh = [torch.randn(H, H, **f) * 0.09 for _ in range(3)]
bs = [torch.randn(H, **f) * 0.1 for _ in range(4)]
sc = torch.rand(H, **f) + 0.5
Pd, Ps = x @ W0[:, H:2 * H].t(), x @ W0[:, 2 * H:].t()
pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(W0.data_ptr(), 3 * H, False, units[0])] + [(Wh[l].data_ptr(), H, False, units[l + 1]) for l in range(3)], dev)
m, e_new = torch.empty(E, H, **f), torch.empty(E, H, **f)
He = [torch.empty(E, H, **f) for _ in range(3)]
Ue, Re = torch.empty(E, H, **f), torch.empty(E, **f)
Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)]
def edge():
    ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, m, He, Ue, Re, ldw0=3 * H,
                adds=[(Pd, topo.dst_s), (Ps, topo.src_s)], wpk=units, saveM=Me)
xn, x_new, mn = torch.randn(N, H, **f), torch.empty(N, H, **f), torch.empty(N, H, **f)
Hn = [torch.empty(N, H, **f) for _ in range(3)]
Un, Rn = torch.empty(N, H, **f), torch.empty(N, **f)
Mn = [torch.empty(N, 4, dtype=torch.int32, device=dev) for _ in range(3)]
agg = torch.randn(N, H, **f)
W0n = torch.randn(H, 2 * H, **f) * 0.05
pkn = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
unitsn = [pkn.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(W0n.data_ptr(), 2 * H, False, unitsn[0])] + [(Wh[l].data_ptr(), H, False, unitsn[l + 1]) for l in range(3)], dev)
def node():
    ops.mlp_fwd(N, H, [(xn, None, H), (agg, None, H)], [W0n] + Wh, bs, sc, H, xn, x_new, mn, Hn, Un, Rn, ldw0=2 * H, wpk=unitsn, saveM=Mn)
dZ = [torch.randn(E, H, **f) for _ in range(4)]
X = [torch.randn(E, H, **f) for _ in range(4)]
gW = [torch.empty(H, 3 * H, **f)] + [torch.empty(H, H, **f) for _ in range(3)]
gb = [torch.empty(H, **f) for _ in range(4)]
nbk = H // 16
def wgrad():
    ops.wgrad([(dZ[0], H, nbk, X[0], H, nbk, H, gW[0], 0, 3 * H, gb[0])] + [(dZ[l], H, nbk, X[l], H, nbk, H, gW[l], 0, H, gb[l]) for l in range(1, 4)], dev)

s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def wall(fa, fb, two, K=40):
    torch.cuda.synchronize()
    best = None
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        for _ in range(K):
            with torch.cuda.stream(s1):
                fa()
            with torch.cuda.stream(s2 if two else s1):
                fb()
            if two:   # the dependency pattern of the backward loop: each stream hands over once per round
                s1.wait_stream(s2)
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        b.record(); b.synchronize()
        t = a.elapsed_time(b) / K
        best = t if best is None else min(best, t)
    return best * 1e3
for f_ in (edge, node, wgrad):
    f_()
torch.cuda.synchronize()
for name, fa in (("edge forward chain", edge), ("node-row chain", node), ("node-row chain x3", lambda: (node(), node(), node()))):
    alone_a, alone_b = wall(fa, lambda: None, False), wall(lambda: None, wgrad, False)
    ser, con = wall(fa, wgrad, False), wall(fa, wgrad, True)
    print(f"{name:20s} alone {alone_a:7.1f} us | wgrad alone {alone_b:7.1f} us | one stream {ser:7.1f} us | two streams {con:7.1f} us  ({100 * (con / ser - 1):+.1f} %)", flush=True)
