"""How the flip-aware gradient criterion of tests/test_hip_configs.py behaves for the training-mode ping-pong instance (MGN_PP=1 / 2)
and for the default kernel on the same cases: differing ReLU masks, distance of the engine and of the fp32 oracle to the fp64
oracle.  usage: python tools/pp_grad_check.py [scan]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import torch
import graph_physics_amd as gp
from conftest import rel_err
from test_hip_configs import _grad_case
dev = torch.device("cuda:0")
def run(g, L, seed, pp):
    if pp: os.environ["MGN_PP"] = pp
    else: os.environ.pop("MGN_PP", None)
    out, grads, (o32, g32), (o64, g64), (flips, total, worst) = _grad_case(dev, g, L, seed)
    h64 = max(rel_err(grads[k], g64[k]) for k in grads); c64 = max(rel_err(g32[k], g64[k]) for k in grads)
    e32 = max(rel_err(grads[k], g32[k]) for k in grads)
    print(f"N={g.x.shape[0]} L={L} seed={seed} MGN_PP={pp or '-'}: flips {flips} of {total} worst {worst:.2e}  engine-vs-fp64 {h64:.3e}  oracle32-vs-fp64 {c64:.3e}  "
          f"engine-vs-oracle32 {e32:.3e}  fwd {rel_err(out, o32):.2e}", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "scan":   # small deep cases: which seeds are flip-free for which kernel
    for seed in range(79, 87):
        g = gp.cylinder_mesh(400, seed - 78)
        for pp in ("", "2"):
            run(g, 15, seed, pp)
else:
    g = gp.cylinder_batch(16, 1885, 0)
    for L, seed in ((3, 94), (3, 95), (2, 96)):
        for pp in ("", "1"):
            run(g, L, seed, pp)
