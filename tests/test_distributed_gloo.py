"""CPU, world_size 2 / 4 / 8, gloo: the N>1 paths.
  * data-parallel replicas: one flat gradient all-reduce per step (bench.py --gpus N);
  * node-partitioned mesh with one-hop halo exchange: forward AND gradients equal the
    un-partitioned oracle (the parity oracle of the partitioned path).
No GPU here, so the partitioned model runs on an injected oracle compute backend -- the
halo / partition / collective logic under test is the product's own."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import recipe as R
from oracle import mgn_oracle as O


class OracleBackend:
    """tests-only compute backend: blocks / MLPs evaluated by the CPU oracle"""

    def prepare(self, edge_index, n_local):
        return edge_index

    @staticmethod
    def _sd(module):
        return {k: v for k, v in module.named_parameters()}

    def mlp(self, module, x):
        return O.mlp(x, self._sd(module), "")

    def order_edges(self, edge_attr, ctx):
        return edge_attr

    def block(self, block, x, e, ctx, pos=None, phi=None):
        variant = {"use_rope": block.use_rope, "rope_axes": block.rope_axes, "rope_base": block.rope_base, "use_gate": block.use_gate}
        return O.graph_net_block(x, e, ctx, self._sd(block), "", variant=variant, pos=pos, phi=phi)

    def transformer_block(self, block, x, edge_index, pos=None):
        a = block.attention
        return O.transformer_block(x, self._sd(block), "", edge_index, a.num_heads, pos=pos, use_rope=a.use_rope_embeddings,
                                   use_gate=a.use_gated_attention, pos_dimension=a.pos_dimension, rope_base=a.rope_base)

    def temporal_block(self, tb, h_prev, h_pred, edge_index):
        return O.temporal_attention(h_prev, h_pred, self._sd(tb), "", edge_index, tb.H, use_gate=tb.use_gate)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_partition(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import graph_physics_amd as gp
    from graph_physics_amd import distributed as D
    from graph_physics_amd import partition as P

    L, H, N, seed = 3, 32, 120, 5
    pos, ei, ea = R.delaunay_graph(N, seed)
    params = R.make_params(R.epd_param_shapes(L, H, 11, 3, 2), seed)
    x_in, e_in = R.randn((N, 11), 1), R.randn((ei.shape[1], 3), 2)
    tgt = R.randn((N, 2), 3)
    nt = torch.from_numpy((np.arange(N) % 3 == 0).astype(np.float32) * 5)  # OUTFLOW / NORMAL mix
    part = P.partition_nodes(pos.numpy(), ei, world)
    plan = P.build_rank_plan(ei, part, rank, world)
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=H)
    net.load_state_dict(params)
    pm = D.PartitionedEPD(net, plan, backend=OracleBackend())
    out = pm(x_in[plan.owned], e_in[plan.edge_ids])
    loss = D.partitioned_loss(out, tgt[plan.owned], nt[plan.owned])
    loss.backward()
    D.GradAllReduce(average=False)(net.parameters())
    grads = {k: v.grad.numpy().copy() for k, v in net.named_parameters()}  # by value: the worker exits first
    q.put((rank, plan.owned.numpy().copy(), out.detach().numpy().copy(), float(loss.detach()), grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_partitioned_forward_backward_equals_unpartitioned(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_partition, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # un-partitioned oracle
    L, H, N, seed = 3, 32, 120, 5
    pos, ei, ea = R.delaunay_graph(N, seed)
    params = {k: v.clone().requires_grad_(True) for k, v in R.make_params(R.epd_param_shapes(L, H, 11, 3, 2), seed).items()}
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
        total += loss  # each rank holds its share of the global masked mean
        for k, g in grads.items():
            assert torch.allclose(torch.from_numpy(g), params[k].grad, rtol=2e-4, atol=1e-6), (rank, k)
    assert abs(total - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
    assert torch.allclose(full, ref.detach(), rtol=1e-5, atol=1e-6)


ROPE_VARIANT = {"use_gate": True, "use_rope": True, "rope_axes": 2, "rope_base": 100.0}


def _worker_partition_rope(rank, world, port, q):
    """RoPE + sigmoid gate with graph.phi on a partitioned mesh: the ghost sources' POSITIONS travel once
    per plan, the ghost latents before every block (per-block path of PartitionedEPD)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import graph_physics_amd as gp
    from graph_physics_amd import distributed as D
    from graph_physics_amd import partition as P

    L, H, N, seed = 2, 32, 140, 15
    pos, ei, ea = R.delaunay_graph(N, seed)
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=H, use_rope_embeddings=True, rope_pos_dimension=2, rope_base=100.0,
                                 use_gated_attention=True)
    net.load_state_dict(R.variant_params(net.state_dict(), seed))
    x_in, e_in = R.randn((N, 11), 11), R.randn((ei.shape[1], 3), 12)
    phi = R.randn((N,), 14)
    tgt = R.randn((N, 2), 13)
    nt = torch.zeros(N)
    part = P.partition_nodes(pos.numpy(), ei, world)
    plan = P.build_rank_plan(ei, part, rank, world)
    pm = D.PartitionedEPD(net, plan, backend=OracleBackend())
    with pytest.raises(ValueError, match="pos"):   # the reference's error when graph.pos is missing (processors.py:188-191)
        pm(x_in[plan.owned], e_in[plan.edge_ids], phi_own=phi[plan.owned])
    out = pm(x_in[plan.owned], e_in[plan.edge_ids], phi_own=phi[plan.owned], pos_own=pos[plan.owned])
    assert pm._pos_full.shape[0] == plan.n_own + plan.n_ghost and torch.equal(pm._pos_full[plan.n_own:], pos[plan.ghost].float())
    loss = D.partitioned_loss(out, tgt[plan.owned], nt[plan.owned])
    loss.backward()
    D.GradAllReduce(average=False)(net.parameters())
    grads = {k: v.grad.numpy().copy() for k, v in net.named_parameters()}
    # [r5, advisor] a deforming mesh: the SAME module called with moved positions must use them -- owned rows and ghosts -- not the
    # rows of its first call; and cache_positions=True keeps the ghosts until invalidate_positions()
    pos2 = pos * 1.5 + 0.25
    with torch.no_grad():
        moved = pm(x_in[plan.owned], e_in[plan.edge_ids], phi_own=phi[plan.owned], pos_own=pos2[plan.owned])
        fresh = D.PartitionedEPD(net, plan, backend=OracleBackend())(x_in[plan.owned], e_in[plan.edge_ids], phi_own=phi[plan.owned],
                                                                     pos_own=pos2[plan.owned])
        assert torch.equal(moved, fresh)
        assert torch.equal(pm._pos_full[plan.n_own:], pos2[plan.ghost].float())
        assert float((moved - out.detach()).abs().max()) > 1e-4     # the positions do matter to this model
        pc = D.PartitionedEPD(net, plan, backend=OracleBackend(), cache_positions=True)
        pc(x_in[plan.owned], e_in[plan.edge_ids], phi_own=phi[plan.owned], pos_own=pos[plan.owned])
        pc(x_in[plan.owned], e_in[plan.edge_ids], phi_own=phi[plan.owned], pos_own=pos2[plan.owned])
        assert torch.equal(pc._pos_full[plan.n_own:], pos[plan.ghost].float())       # cached ghosts, as promised by the caller
        assert torch.equal(pc._pos_full[:plan.n_own], pos2[plan.owned].float())      # owned rows always follow the argument
        pc.invalidate_positions()
        again = pc(x_in[plan.owned], e_in[plan.edge_ids], phi_own=phi[plan.owned], pos_own=pos2[plan.owned])
        assert torch.equal(again, fresh)
    q.put((rank, plan.owned.numpy().copy(), out.detach().numpy().copy(), float(loss.detach()), grads))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_rope_and_gate_phi_equal_unpartitioned_world4():
    world = 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_partition_rope, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import graph_physics_amd as gp

    L, H, N, seed = 2, 32, 140, 15
    pos, ei, ea = R.delaunay_graph(N, seed)
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=H, use_rope_embeddings=True, rope_pos_dimension=2, rope_base=100.0,
                                 use_gated_attention=True)
    params = {k: v.clone().requires_grad_(True) for k, v in R.variant_params(net.state_dict(), seed).items()}
    x_in, e_in = R.randn((N, 11), 11), R.randn((ei.shape[1], 3), 12)
    phi, tgt, nt = R.randn((N,), 14), R.randn((N, 2), 13), torch.zeros(N)
    ref = O.epd_forward(x_in, e_in, ei, params, L, variant=ROPE_VARIANT, pos=pos, phi=phi)
    ref_loss = O.l2_loss(ref, tgt, nt)
    ref_loss.backward()
    full = torch.zeros_like(ref)
    total = 0.0
    for rank, owned, out, loss, grads in res:
        full[torch.from_numpy(owned)] = torch.from_numpy(out)
        total += loss
        for k, g in grads.items():
            assert torch.allclose(torch.from_numpy(g), params[k].grad, rtol=2e-4, atol=1e-6), (rank, k)
    assert abs(total - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
    assert torch.allclose(full, ref.detach(), rtol=1e-5, atol=1e-6)


ETD_CASE = dict(L=3, H=32, heads=4, N=160, seed=31)


def _etd_net(gp, rope, temporal=False):
    c = ETD_CASE
    return gp.EncodeTransformDecode(c["L"], 11, 2, hidden_size=c["H"], num_heads=c["heads"], use_rope_embeddings=rope,
                                    use_gated_attention=rope, rope_pos_dimension=2, rope_base=100.0, use_temporal_block=temporal)


def _worker_partition_etd(rank, world, port, q, rope, temporal=False):
    """[r5] the sparse-attention Transformer on a partitioned mesh (VERDICT r4 missing 5): a rank owns the attention ROWS
    (edge_index[0]) of its nodes -- the plan is built on the flipped edge list -- and receives the ghost columns' latents before
    every block; with RoPE also their positions"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import graph_physics_amd as gp
    from graph_physics_amd import distributed as D
    from graph_physics_amd import partition as P

    c = ETD_CASE
    pos, ei, _ = R.delaunay_graph(c["N"], c["seed"])
    net = _etd_net(gp, rope, temporal)
    net.load_state_dict(R.variant_params(net.state_dict(), c["seed"]))
    x_in, tgt, nt = R.randn((c["N"], 11), 41), R.randn((c["N"], 2), 43), torch.zeros(c["N"])
    part = P.partition_nodes(pos.numpy(), ei, world)
    plan = P.build_rank_plan(ei.flip(0), part, rank, world)          # owner of an edge = owner of its attention row
    pm = D.PartitionedETD(net, plan, backend=OracleBackend())
    lei = pm.local_edge_index()
    assert int(lei[0].max()) < plan.n_own                             # every local row is an owned node
    if rope:
        with pytest.raises(ValueError, match="pos"):
            pm(x_in[plan.owned])
    out = pm(x_in[plan.owned], pos_own=pos[plan.owned] if rope else None)
    loss = D.partitioned_loss(out, tgt[plan.owned], nt[plan.owned])
    loss.backward()
    D.GradAllReduce(average=False)(net.parameters())
    grads = {k: v.grad.numpy().copy() for k, v in net.named_parameters()}
    q.put((rank, plan.owned.numpy().copy(), out.detach().numpy().copy(), float(loss.detach()), grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rope,temporal", [(False, False), (True, False), (False, True)])
def test_partitioned_transformer_equals_unpartitioned_world4(rope, temporal):
    """temporal [r6]: the temporal block (processors.py:376-377) after the last Transformer block -- prev_x with its ghosts is the
    last block's exchanged input, the ghosts of last_x travel in one more exchange"""
    world = 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_partition_etd, args=(r, world, port, q, rope, temporal)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import graph_physics_amd as gp

    c = ETD_CASE
    pos, ei, _ = R.delaunay_graph(c["N"], c["seed"])
    net = _etd_net(gp, rope, temporal)
    params = {k: v.clone().requires_grad_(True) for k, v in R.variant_params(net.state_dict(), c["seed"]).items()}
    x_in, tgt, nt = R.randn((c["N"], 11), 41), R.randn((c["N"], 2), 43), torch.zeros(c["N"])
    ref = O.etd_forward(x_in, ei, params, c["L"], c["heads"], pos=pos if rope else None, use_rope=rope, use_gate=rope,
                        pos_dimension=2, rope_base=100.0, use_temporal_block=temporal)
    ref_loss = O.l2_loss(ref, tgt, nt)
    ref_loss.backward()
    full, total = torch.zeros_like(ref), 0.0
    for rank, owned, out, loss, grads in res:
        full[torch.from_numpy(owned)] = torch.from_numpy(out)
        total += loss
        for k, g in grads.items():
            if params[k].grad is None:      # (buffers / unused entries)
                continue
            assert torch.allclose(torch.from_numpy(g), params[k].grad, rtol=2e-4, atol=1e-6), (rank, k)
    assert abs(total - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
    assert torch.allclose(full, ref.detach(), rtol=1e-5, atol=1e-6)


EPD_T_CASE = dict(L=3, H=32, N=140, seed=71)


def _epd_t_graph():
    """a DIRECTED mesh: a third of the Delaunay edges lose their reverse, so the attention rows (sources) and the
    message-passing destinations of a rank have different ghosts and a different interior / boundary split"""
    c = EPD_T_CASE
    pos, ei, _ = R.delaunay_graph(c["N"], c["seed"])
    keep = torch.from_numpy(np.random.default_rng(c["seed"]).random(ei.shape[1]) > 0.33) | (ei[0] < ei[1])
    return pos, ei[:, keep].contiguous()


def _worker_partition_epd_temporal(rank, world, port, q, backend):
    """[r6] EncodeProcessDecode(use_temporal_block=True) on a partitioned mesh: message passing on the plan of the edge list, the
    temporal block (rows = SOURCES, processors.py:183-184) on the plan of the flipped one, same partition vector"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import graph_physics_amd as gp
    from graph_physics_amd import distributed as D
    from graph_physics_amd import partition as P

    c = EPD_T_CASE
    pos, ei = _epd_t_graph()
    net = gp.EncodeProcessDecode(c["L"], 11, 3, 2, hidden_size=c["H"], use_temporal_block=True, attention_backend=backend)
    net.load_state_dict(R.variant_params(net.state_dict(), c["seed"]))
    x_in, e_in, tgt, nt = R.randn((c["N"], 11), 72), R.randn((ei.shape[1], 3), 73), R.randn((c["N"], 2), 74), torch.zeros(c["N"])
    part = P.partition_nodes(pos.numpy(), ei, world)
    plan = P.build_rank_plan(ei, part, rank, world)
    if backend == "dgl":
        with pytest.raises(ValueError, match="temporal_plan"):
            D.PartitionedEPD(net, plan, backend=OracleBackend())
        tplan = P.build_rank_plan(ei.flip(0), part, rank, world)
    else:
        tplan = None                                                   # no adjacency: the block is row-wise
    pm = D.PartitionedEPD(net, plan, backend=OracleBackend(), temporal_plan=tplan)
    out = pm(x_in[plan.owned], e_in[plan.edge_ids])
    loss = D.partitioned_loss(out, tgt[plan.owned], nt[plan.owned])
    loss.backward()
    D.GradAllReduce(average=False)(net.parameters())
    grads = {k: v.grad.numpy().copy() for k, v in net.named_parameters()}
    q.put((rank, plan.owned.numpy().copy(), out.detach().numpy().copy(), float(loss.detach()), grads, pm._t_of is not None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["dgl", "pyg"])
def test_partitioned_epd_temporal_block_equals_unpartitioned_world4(backend):
    world = 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_partition_epd_temporal, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import graph_physics_amd as gp

    c = EPD_T_CASE
    pos, ei = _epd_t_graph()
    net = gp.EncodeProcessDecode(c["L"], 11, 3, 2, hidden_size=c["H"], use_temporal_block=True, attention_backend=backend)
    p_ = {k: v.clone().requires_grad_(True) for k, v in R.variant_params(net.state_dict(), c["seed"]).items()}
    x_in, e_in, tgt, nt = R.randn((c["N"], 11), 72), R.randn((ei.shape[1], 3), 73), R.randn((c["N"], 2), 74), torch.zeros(c["N"])
    x, e = O.mlp(x_in, p_, "nodes_encoder."), O.mlp(e_in, p_, "edges_encoder.")
    prev = x
    for i in range(c["L"]):
        prev = x
        x, e = O.graph_net_block(x, e, ei, p_, f"processor_list.{i}.")
    ref = O.mlp(O.temporal_attention(prev, x, p_, "temporal_block.", ei if backend == "dgl" else None, 4), p_, "decode_module.")
    ref_loss = O.l2_loss(ref, tgt, nt)
    ref_loss.backward()
    full, total = torch.zeros_like(ref), 0.0
    for rank, owned, out, loss, grads, renumbered in res:
        full[torch.from_numpy(owned)] = torch.from_numpy(out)
        total += loss
        for k, g in grads.items():
            if p_[k].grad is None:
                continue
            # (summation order differs between 4 partial sums and one: absolute slack scaled by the gradient's size)
            assert torch.allclose(torch.from_numpy(g), p_[k].grad, rtol=2e-4, atol=1e-6 * max(1.0, float(p_[k].grad.abs().max()))), (rank, k)
    if backend == "dgl":
        assert any(r[5] for r in res)      # the directed mesh did exercise the owned-row renumbering between the two plans
    assert abs(total - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
    assert torch.allclose(full, ref.detach(), rtol=1e-5, atol=1e-6)


def test_rcb_partition_and_plan():
    from graph_physics_amd import partition as P

    pos, ei, _ = R.delaunay_graph(1000, 9)
    for k in (2, 3, 8):
        part = P.rcb_partition(pos.numpy(), k)
        sizes = np.bincount(part, minlength=k)
        assert sizes.max() - sizes.min() <= 1
        assert P.edge_cut(ei, part) < 0.25
        plans = [P.build_rank_plan(ei, part, r, k) for r in range(k)]
        src, dst = ei[0].numpy(), ei[1].numpy()
        for p in plans:
            # interior nodes first: none of their in-edges has a remote source; every boundary node has one
            remote_in = np.zeros(1000, dtype=bool)
            mine = part[dst] == p.rank
            remote_in[dst[mine & (part[src] != p.rank)]] = True
            own = p.owned.numpy()
            assert not remote_in[own[:p.n_interior]].any() and remote_in[own[p.n_interior:]].all()
            assert p.n_interior_edges == int(np.count_nonzero(~remote_in[dst[mine]]))
            # the send list grouped by node covers every send row exactly once, in ascending position
            sp, rp, sn = p.send_perm.numpy(), p.send_rowptr.numpy(), p.send_nodes.numpy()
            assert np.array_equal(np.sort(sp), np.arange(p.send_idx.numel()))
            for j in range(sn.size):
                seg = sp[rp[j]:rp[j + 1]]
                assert (p.send_idx.numpy()[seg] == sn[j]).all() and (np.diff(seg) > 0).all()
        assert sum(p.edge_ids.numel() for p in plans) == ei.shape[1]  # every edge lives on exactly one rank
        assert sum(p.n_own for p in plans) == 1000
        for p in plans:
            assert int(p.edge_index[1].max()) < p.n_own  # destinations are owned
            assert sum(p.recv_counts) == p.n_ghost and sum(p.send_counts) == p.send_idx.numel()
            for qq in range(k):  # what I send to q is what q expects from me, in the same order
                if qq == p.rank:
                    continue
                a = sum(p.send_counts[:qq])
                mine = p.owned[p.send_idx[a:a + p.send_counts[qq]]]
                b = sum(plans[qq].recv_counts[:p.rank])
                theirs = plans[qq].ghost[b:b + plans[qq].recv_counts[p.rank]]
                assert torch.equal(mine, theirs)


def _worker_dp(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graph_physics_amd import distributed as D

    torch.manual_seed(rank)
    lin = torch.nn.Linear(5, 3)
    D.broadcast_parameters(lin)
    w0 = lin.weight.detach().clone()
    x = torch.full((4, 5), float(rank + 1))
    lin(x).sum().backward()
    D.GradAllReduce()(lin.parameters())
    q.put((rank, w0.numpy().copy(), lin.weight.grad.numpy().copy(), lin.bias.grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_grad_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_dp, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1])  # broadcast: same start weights
    # grad wrt W of sum(lin(x)) = column sums of x: rank r gives 4*(r+1); average = 6
    assert np.allclose(res[0][2], 6.0) and np.array_equal(res[0][2], res[1][2])
    assert np.allclose(res[0][3], 4.0)


def test_graph_partitioners_cut_and_balance():
    """refinement never worsens the RCB cut; the multilevel (METIS-style, graph-only) partitioner beats
    coordinate bisection where coordinates mislead (a spiral tube), within its balance bound."""
    from graph_physics_amd import partition as P
    from graph_physics_amd.mesh import faces_to_edges
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(0)
    n = 12000
    t = rng.random(n) * 3 * np.pi
    r, ph = 0.1 * np.sqrt(rng.random(n)), rng.random(n) * 2 * np.pi
    c = np.stack([np.cos(t) * (1 + 0.3 * t), np.sin(t) * (1 + 0.3 * t), 0.2 * t], 1)
    p3 = c + np.stack([r * np.cos(ph), r * np.sin(ph), r * np.sin(ph + 1)], 1)
    ei = faces_to_edges(Delaunay(p3).simplices, n)
    keep = np.linalg.norm(p3[ei[0]] - p3[ei[1]], axis=1) < 0.12   # the tube's own edges only
    ei = torch.from_numpy(ei[:, keep])
    for k in (4, 8):
        rcb = P.rcb_partition(p3, k)
        ref = P.refine_partition(ei, rcb, k)
        ml = P.partition_nodes(None, ei, k)          # no coordinates: multilevel
        assert P.edge_cut(ei, ref) <= P.edge_cut(ei, rcb)
        assert P.edge_cut(ei, ml) < 0.8 * P.edge_cut(ei, rcb), (P.edge_cut(ei, ml), P.edge_cut(ei, rcb))
        assert P.imbalance_of(ml, k) < 0.06 and P.imbalance_of(ref, k) < 0.03
        assert set(np.unique(ml)) == set(range(k))
