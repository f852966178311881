"""worker of tests/test_halo_rccl_single.py: ONE rank over RCCL, self-exchange halo plan (see the test's docstring)"""
import os, sys
import numpy as np, torch, torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
import recipe as R, graph_physics_amd as gp
from graph_physics_amd import distributed as D, partition as P
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
dev = torch.device("cuda:0")
L, N, seed = 4, 3000, 17
pos, ei, ea = R.delaunay_graph(N, seed)
params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), seed)
x_in, e_in, cot = R.randn((N, 11), 1).to(dev), R.randn((ei.shape[1], 3), 2).to(dev), R.randn((N, 2), 3).to(dev)
# reference: the plain engine
ref = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev); ref.load_state_dict(params)
out_ref = ref(gp.Graph(x=x_in, edge_attr=e_in, edge_index=ei.to(dev)))
(out_ref * cot).sum().backward()
# self-exchange plan: sources in B become ghosts for a third of their out-edges
rng = np.random.default_rng(0)
src, dst = ei[0].numpy(), ei[1].numpy()
B = np.sort(rng.choice(N, 400, replace=False))
pos_in_B = np.full(N, -1); pos_in_B[B] = np.arange(B.size)
redirect = (pos_in_B[src] >= 0) & (rng.random(src.size) < 0.35)
is_bnd = np.zeros(N, dtype=bool); is_bnd[dst[redirect]] = True
owned = np.concatenate([np.nonzero(~is_bnd)[0], np.nonzero(is_bnd)[0]])      # interior first, then boundary
loc = np.empty(N, dtype=np.int64); loc[owned] = np.arange(N)
lsrc = np.where(redirect, N + pos_in_B[src], loc[src])
le = torch.from_numpy(np.stack([lsrc, loc[dst]]))
send_idx = loc[B]
perm = np.argsort(send_idx, kind="stable"); uniq, counts = np.unique(send_idx[perm], return_counts=True)
rowptr = np.zeros(uniq.size + 1, dtype=np.int64); np.cumsum(counts, out=rowptr[1:])
plan = P.RankPlan(0, 1, torch.from_numpy(owned), torch.from_numpy(B.copy()), torch.arange(ei.shape[1]), le,
                  torch.from_numpy(send_idx), [int(B.size)], [int(B.size)], n_interior=int((~is_bnd).sum()),
                  n_interior_edges=int(np.count_nonzero(~is_bnd[dst])), send_nodes=torch.from_numpy(uniq),
                  send_rowptr=torch.from_numpy(rowptr), send_perm=torch.from_numpy(perm))
plan.self_exchange = True
assert 0 < plan.n_interior < N and 0 < plan.n_interior_edges < ei.shape[1]
net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev); net.load_state_dict(params)
pm = D.PartitionedEPD(net, plan)
own = torch.from_numpy(owned).to(dev)
runs = []
for _ in range(2):
    net.zero_grad(set_to_none=True)
    out = pm(x_in[own], e_in)
    (out * cot[own]).sum().backward()
    runs.append((out.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
assert pm._halo is not None and pm._halo.active, "the exchange did not go through the process group"
torch.cuda.synchronize()
err = float((runs[0][0] - out_ref.detach()[own]).abs().max() / out_ref.detach().abs().max())
assert err < 2e-6, err
for (k, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
    ge = float((runs[0][1][k] - b.grad).abs().max() / b.grad.abs().max())
    assert ge < 1e-3, (k, ge)   # other summation order of the scattered gradients: rounding, then a ReLU mask or two (see test_hip_configs)
    assert torch.equal(runs[0][1][k], runs[1][1][k]), k          # fixed-order unpack: bit-reproducible
dist.destroy_process_group()
print("ok", err)
