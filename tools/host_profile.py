#!/usr/bin/env python3
"""Where the HOST time of one training step goes (cProfile over enqueue-only steps: no synchronisation inside the profiled
region).  python tools/host_profile.py [partitioned]"""
import cProfile, os, pstats, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch, graph_physics_amd as gp
from graph_physics_amd import harness, distributed as D, partition as P
dev = torch.device("cuda:0")
part = len(sys.argv) > 1 and sys.argv[1] == "partitioned"
nb = int(os.environ.get("HP_BATCH", "16"))
g = gp.cylinder_batch(nb, 1885, 0)
net = gp.EncodeProcessDecode(15, g.x.shape[1], g.edge_attr.shape[1], 2, hidden_size=128).to(dev)
tgt = torch.randn(g.x.shape[0], 2)
nt = torch.zeros(g.x.shape[0])
opt = harness.FusedClipAdamW(net.parameters(), 1e-4, max_norm=1.0)
if part:
    pr = P.partition_nodes(g.pos.numpy(), g.edge_index, 8)
    plan = P.build_rank_plan(g.edge_index, pr, 0, 8, pos=g.pos.numpy())
    plan.world = 1
    pm = D.PartitionedEPD(net, plan)
    xo, eo, to, no = g.x[plan.owned].to(dev), g.edge_attr[plan.edge_ids].to(dev), tgt[plan.owned].to(dev), nt[plan.owned].to(dev)
    fwd = lambda: pm(xo, eo)   # noqa: E731
else:
    gd = g.to(dev)
    to, no = tgt.to(dev), nt.to(dev)
    fwd = lambda: net(gd)      # noqa: E731


def step():
    loss = harness.l2_loss(fwd(), to, no)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    step()
dt = (time.perf_counter() - t0) / K
torch.cuda.synchronize()
print("host enqueue time per step: %.2f ms (%s, batch %d)" % (dt * 1e3, "rank share 1/8" if part else "whole batch", nb))
pr_ = cProfile.Profile()
pr_.enable()
for _ in range(K):
    step()
pr_.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr_)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)

# ---- the backward pass runs on autograd's device thread (cProfile does not see it): wall time inside the engine's launch functions
# against the whole of ProcessorFunction.backward
import collections
from graph_physics_amd import ops as _ops
acc = collections.defaultdict(lambda: [0, 0.0])


def _wrap(name):
    fn = getattr(_ops, name)

    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        v = acc[name]
        v[0] += 1
        v[1] += time.perf_counter() - t
        return r
    setattr(_ops, name, w)


for n_ in ("mlp_fwd", "mlp_bwd", "wgrad", "segsum2", "segsum_topo", "seg_fix", "colred_batch", "wpack", "_split_block", "gather_rows"):
    _wrap(n_)
_bw = _ops.ProcessorFunction.backward


def _bw_t(ctx, *g):
    t = time.perf_counter()
    r = _bw(ctx, *g)
    v = acc["ProcessorFunction.backward (total)"]
    v[0] += 1
    v[1] += time.perf_counter() - t
    return r


_ops.ProcessorFunction.backward = staticmethod(_bw_t)
_emp = torch.empty
_el = torch.empty_like


def _empty(*a, **k):
    t = time.perf_counter()
    r = _emp(*a, **k)
    v = acc["torch.empty"]
    v[0] += 1
    v[1] += time.perf_counter() - t
    return r


torch.empty = _empty
for _ in range(K):
    step()
torch.cuda.synchronize()
torch.empty = _emp
print("per step (host wall time inside, %d steps):" % K)
for k_, (n_, t_) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("  %-40s %6.1f calls  %7.3f ms" % (k_, n_ / K, 1e3 * t_ / K))
