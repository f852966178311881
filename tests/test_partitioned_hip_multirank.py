"""GPU, world_size 2: the node-partitioned mesh on the HIP backend -- real kernels AND a real halo
exchange.  A 1-GPU box cannot run RCCL with two ranks on one device, so both ranks share cuda:0
and the collectives go over gloo (the all_to_all_single of the halo exchange falls back to
point-to-point copies through the host there; the partition / halo / gradient-sum logic and the
kernels are the product's).  Forward, loss and every weight gradient must equal the
un-partitioned oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import recipe as R
from oracle import mgn_oracle as O

pytestmark = pytest.mark.gpu

L, H, N, SEED = 3, 128, 900, 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import graph_physics_amd as gp
    from graph_physics_amd import distributed as D
    from graph_physics_amd import partition as P

    dev = torch.device("cuda:0")
    pos, ei, ea = R.delaunay_graph(N, SEED)
    params = R.make_params(R.epd_param_shapes(L, H, 11, 3, 2), SEED)
    x_in, e_in = R.randn((N, 11), 1), R.randn((ei.shape[1], 3), 2)
    tgt = R.randn((N, 2), 3)
    nt = torch.from_numpy((np.arange(N) % 3 == 0).astype(np.float32) * 5)
    part = P.partition_nodes(pos.numpy(), ei, world)
    plan = P.build_rank_plan(ei, part, rank, world)
    assert plan.n_ghost > 0 and 0 < plan.n_interior < plan.n_own and 0 < plan.n_interior_edges < plan.edge_ids.numel()
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=H).to(dev)
    net.load_state_dict(params)
    pm = D.PartitionedEPD(net, plan)  # default backend: the HIP engine, halo exchange inside the processor node
    runs = []
    for _ in range(2):  # twice: the partitioned gradients are bit-reproducible (no atomics on the path)
        net.zero_grad(set_to_none=True)
        out = pm(x_in[plan.owned].to(dev), e_in[plan.edge_ids].to(dev))
        loss = D.partitioned_loss(out, tgt[plan.owned].to(dev), nt[plan.owned].to(dev))
        loss.backward()
        runs.append({k: v.grad.clone() for k, v in net.named_parameters()})
    assert pm._halo is not None and pm._halo.active
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
    D.GradAllReduce(average=False)(net.parameters())
    grads = {k: v.grad.cpu().numpy().copy() for k, v in net.named_parameters()}
    q.put((rank, plan.owned.numpy().copy(), out.detach().cpu().numpy().copy(), float(loss.detach()), grads))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_hip_two_ranks_equal_unpartitioned_oracle():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    pos, ei, ea = R.delaunay_graph(N, SEED)
    params = {k: v.clone().requires_grad_(True) for k, v in R.make_params(R.epd_param_shapes(L, H, 11, 3, 2), SEED).items()}
    x_in, e_in = R.randn((N, 11), 1), R.randn((ei.shape[1], 3), 2)
    tgt = R.randn((N, 2), 3)
    nt = torch.from_numpy((np.arange(N) % 3 == 0).astype(np.float32) * 5)
    ref = O.epd_forward(x_in, e_in, ei, params, L)
    ref_loss = O.l2_loss(ref, tgt, nt)
    ref_loss.backward()
    full = torch.zeros_like(ref)
    total = 0.0
    for rank, owned, out, loss, grads in res:
        full[torch.from_numpy(owned)] = torch.from_numpy(out)
        total += loss
        for k, g in grads.items():
            gref = params[k].grad
            err = float((torch.from_numpy(g) - gref).abs().max() / gref.abs().max())
            assert err < 3e-4, (rank, k, err)  # the suite's 1e-4-per-round gradient criterion
    assert abs(total - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
    assert float((full - ref.detach()).abs().max() / ref.detach().abs().max()) < 1e-5
