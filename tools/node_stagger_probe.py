#!/usr/bin/env python3
"""Experiment: the node-update launch of the bench shape (N = 30 160 rows, 7 GEMM units, training saves) with the second
dispatch round (workgroups >= 256 share their CUs with the first) started `t` x 10 ns late, so that the two waves of a SIMD are out
of phase and one's boundary work runs under the other's MFMAs.  Needs tools/libexp_STG.so (-DMGN_EXP_PAIR_STAGGER).
python tools/node_stagger_probe.py 0 100 200 400"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["MGN_LIB"] = os.path.join(R, "tools", os.environ.get("TL_LIB", "libexp_STG.so"))
import torch, graph_physics_amd as gp
from graph_physics_amd import ops, _capi
dev = torch.device("cuda:0")
nb = int(os.environ.get("TL_BATCH", "16"))
g = gp.cylinder_batch(nb, 1885, 0).to(dev)
N, H = g.x.shape[0], 128
f = dict(dtype=torch.float32, device=dev)
x, agg = torch.randn(N, H, **f), torch.randn(N, H, **f)
W0 = torch.randn(H, 3 * H, **f) * 0.05
Wn0 = torch.randn(H, 2 * H, **f) * 0.05
Wh = [torch.randn(H, H, **f) * 0.09 for _ in range(3)]
bs = [torch.zeros(H, **f) for _ in range(4)]
sc = torch.ones(H, **f)
pkn = torch.empty(7 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
un = [pkn.data_ptr() + u * _capi.WPACK_BYTES for u in range(7)]
ops.wpack([(Wn0.data_ptr(), 2 * H, False, un[0]), (Wn0.data_ptr() + 4 * H, 2 * H, False, un[1])] +
          [(Wh[l].data_ptr(), H, False, un[2 + l]) for l in range(3)] +
          [(W0.data_ptr() + 4 * H, 3 * H, False, un[5]), (W0.data_ptr() + 8 * H, 3 * H, False, un[6])], dev)
sets = []
for _ in range(4):   # rotating outputs: the stores of a launch do not hit lines the previous one left in the cache
    sets.append(dict(x_new=torch.empty(N, H, **f), Hn=[torch.empty(N, H, **f) for _ in range(3)], Un=torch.empty(N, H, **f),
                     Rn=torch.empty(N, **f), Pd=torch.empty(N, H, **f), Ps=torch.empty(N, H, **f)))
L = _capi.lib()
save = os.environ.get("TL_SAVE", "1") == "1"


def launch(s):
    ops.mlp_fwd(N, H, [(x, None, H), (agg, None, H)], [Wn0] + Wh, bs, sc, H, x, s["x_new"], None, s["Hn"] if save else None,
                s["Un"] if save else None, s["Rn"] if save else None,
                posts=[(W0.data_ptr() + 4 * H, s["Pd"]), (W0.data_ptr() + 8 * H, s["Ps"])], post_ldw=3 * H, wpk=un)


for t in [int(v) for v in sys.argv[1:]] or [0, 100, 200, 400]:
    if hasattr(L, "mgn_debug_set_stagger"):
        assert L.mgn_debug_set_stagger(t) == 0
    for i in range(20):
        launch(sets[i % 4])
    torch.cuda.synchronize()
    best = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            launch(sets[i % 4])
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 10)
    chk = [float(sets[0][k].double().abs().sum()) for k in ("x_new", "Pd", "Ps")] + ([float(sets[0]["Un"].double().abs().sum())] if save else [])
    print("stagger %4d ticks (save=%d, N=%d, MGN_NW_SMALL=%s): %.1f us per launch (min of 5 x 100), median %.1f   checksums %s" % (
        t, save, N, os.environ.get("MGN_NW_SMALL", "-"), min(best), sorted(best)[2], " ".join("%.9e" % v for v in chk)), flush=True)
