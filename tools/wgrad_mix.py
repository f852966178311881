import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, graph_physics_amd as gp
from graph_physics_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda:0")
g = gp.cylinder_batch(16, 1885, 0).to(dev)
topo = ops.Topology(g.edge_index, g.x.shape[0])
N, E, H = topo.N, topo.E, 128
f = dict(dtype=torch.float32, device=dev)
rE = lambda: torch.randn(E, H, **f)
rN = lambda: torch.randn(N, H, **f)
nb = 8
def job(A, B): return (A, H, nb, B, H, nb, H, torch.empty(H, H, **f), 0, H)
ej = [job(rE(), rE()) for _ in range(4)]
nj = [job(rN(), rN()) for _ in range(6)]
for name, jobs in (("4 E-jobs", ej), ("4 E + 2 N", ej + nj[:2]), ("4 E + 6 N", ej + nj), ("6 N-jobs", nj), ("1 E-job", ej[:1]), ("2 E-jobs", ej[:2])):
    fl = 2.0 * H * H * sum(j[0].shape[0] for j in jobs)
    t = timeit(lambda: ops.wgrad(jobs, dev))
    print(f"{name:12s} {t*1e3:8.1f} us  {fl/t/1e9:6.1f} TFLOP/s")
