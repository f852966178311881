#!/usr/bin/env python3
"""bf16 matrix mode (one term per product), edge-update shape without the fused aggregation: does the weight ring's depth bound the
chain kernel there?  The same launch on the 4-wave instance (three 24 KB ring slots, two quarters ahead) and the 8-wave instance
(six slots, five ahead), dynamic shapes, next to the six-term launch.  usage: MGN_NW=4|8 python tools/kbench_b16_ring.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_physics_amd import ops, _capi
from tools.kbench import timeit
dev = torch.device("cuda:0")
E, N, H = int(sys.argv[1]) if len(sys.argv) > 1 else 338842, 20800, 128
f = dict(dtype=torch.float32, device=dev)
torch.manual_seed(0)
x, e = torch.randn(N, H, **f), torch.randn(E, H, **f)
dst = torch.sort(torch.randint(0, N, (E,), device=dev, dtype=torch.int32)).values
src = torch.randint(0, N, (E,), device=dev, dtype=torch.int32)
W0 = torch.randn(H, 3 * H, **f) * 0.05
Wh = [torch.randn(H, H, **f) * 0.09 for _ in range(3)]
bs = [torch.randn(H, **f) * 0.1 for _ in range(4)]
sc = torch.rand(H, **f) + 0.5
Pd, Ps = x @ W0[:, H:2 * H].t(), x @ W0[:, 2 * H:].t()
pk = torch.empty(4 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
units = [pk.data_ptr() + u * _capi.WPACK_BYTES for u in range(4)]
ops.wpack([(W0.data_ptr(), 3 * H, False, units[0])] + [(Wh[l].data_ptr(), H, False, units[l + 1]) for l in range(3)], dev)
m, e_new = torch.empty(E, H, **f), torch.empty(E, H, **f)
for prec, save in ((1, False), (1, True), (0, True)):
    He = [torch.empty(E, H, **f) for _ in range(3)] if save else None
    Ue, Re = (torch.empty(E, H, **f), torch.empty(E, **f)) if save else (None, None)
    Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)] if save else None
    fn = lambda: ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, m, He, Ue, Re, ldw0=3 * H,  # noqa: E731
                             adds=[(Pd, dst), (Ps, src)], wpk=units, saveM=Me, precision=prec)
    t = timeit(fn)
    print(f"MGN_NW={os.environ.get('MGN_NW', 'default')} MGN_X6_STATIC={os.environ.get('MGN_X6_STATIC', 'default')}  "
          f"{'bf16 (1 term)' if prec else 'fp32 (6 terms)'} saves={int(save)}: {t * 1e3:7.1f} us for {E} rows", flush=True)
