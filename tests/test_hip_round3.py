"""GPU: what the round-2 review asked for.

  * configs[3] AS STATED: the 1M-node mesh numbered the way its generator drew the points (no locality); the
    engine renumbers on entry (Morton order of the positions, on the device) and un-does it on exit -- forward
    against the oracle on L-hop closures, renumbered == raw, and GRADIENT parity at that size: a cotangent
    supported on seed nodes makes every parameter gradient a function of the L-hop closure only, so the oracle
    on the closure sub-mesh is exact (activation recompute on and off);
  * the topology build without a host synchronisation (lazy flags): arrays identical to the eager build, a stray
    index is reported at the next wait and nothing faults before, hub nodes on the optimistic first pass;
  * node renumbering (Morton with positions, reverse Cuthill-McKee without) against the oracle, forward and
    gradients;
  * noise injection with overlapping column ranges (the reference applies them one after the other).
"""
import functools

import numpy as np
import pytest
import torch

import recipe as R
import graph_physics_amd as gp
from conftest import assert_close3, rel_err
from graph_physics_amd import ops
from oracle import mgn_oracle as O

pytestmark = pytest.mark.gpu
FWD_TOL = 1e-5
GRAD_TOL = 1e-4


class silu_models:
    """Gradient bars against the fp32 oracle are taken on SiLU networks (the reference's ``use_silu_activation`` switch,
    layers.py:132-160): with ReLU a pre-activation at rounding distance from zero may take the other branch on the
    other machine, which moves BOTH sides ~1e-3 from an fp64 evaluation (tests/test_hip_configs.py::_check_grads
    measures and bounds that); SiLU has no branch, so the comparison stays at rounding level."""

    def __enter__(self):
        from graph_physics_amd import layers
        self._layers, self._old = layers, layers.use_silu_activation()
        layers.set_use_silu_activation(True)

    def __exit__(self, *a):
        self._layers.set_use_silu_activation(self._old)


# ------------------------------------------------------------------ Morton order
def _morton_keys_np(pos, D):
    bits = 31 if D == 2 else 21
    p = pos[:, :D].astype(np.float32)
    lo, hi = p.min(axis=0), p.max(axis=0)
    t = (p.astype(np.float64) - lo.astype(np.float64)) / (hi.astype(np.float64) - lo.astype(np.float64))
    q = (np.clip(t, 0.0, 1.0) * float((1 << bits) - 1)).astype(np.uint64)
    key = np.zeros(p.shape[0], dtype=np.uint64)
    for b in range(bits):
        for a in range(D):
            key |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * D + a)
    return key


@pytest.mark.parametrize("D,n", [(2, 5000), (3, 4097), (2, 1)])
def test_morton_order_is_the_sorted_key_order(dev, D, n):
    rng = np.random.default_rng(3 + D)
    pos = (rng.random((n, D + 1)) * np.array([3.0, 0.5, 2.0, 1.0])[: D + 1] - 0.7).astype(np.float32)
    order, rank = ops.morton_order(torch.from_numpy(pos[:, :D].copy()).to(dev))
    order, rank = order.cpu().numpy().astype(np.int64), rank.cpu().numpy().astype(np.int64)
    assert sorted(order.tolist()) == list(range(n))
    assert np.array_equal(rank[order], np.arange(n))
    key = _morton_keys_np(pos, D) if n > 1 else np.zeros(n, dtype=np.uint64)
    want = np.argsort(key, kind="stable")            # ties by old id: the radix sort is stable
    assert np.array_equal(order, want)


# ------------------------------------------------------------------ lazy topology build
def test_lazy_topology_equals_eager_and_reports_stray_indices_late(dev):
    ei = R.random_graph(300, 4000, 7)
    a = ops.Topology(ei.to(dev), 300)
    b = ops.Topology(ei.to(dev), 300, lazy=True)
    assert not b.resolved and b.max_in_degree is None
    for f in ("rowptr_dst", "perm_dst", "src_s", "dst_s", "rowptr_src", "perm_src"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    b.resolve()
    assert b.resolved and (b.max_in_degree, b.max_out_degree) == (a.max_in_degree, a.max_out_degree)
    # a stray index: the eager build raises at once, the lazy one at resolve(); in between every array is safe to
    # compute on (an EncodeProcessDecode forward runs to the end and THEN raises)
    bad = ei.clone()
    bad[0, 17] = 300
    bad[1, 900] = -2
    with pytest.raises(IndexError):
        ops.Topology(bad.to(dev), 300)
    t = ops.Topology(bad.to(dev), 300, lazy=True)
    assert int(t.src_s.min()) >= 0 and int(t.src_s.max()) < 300 and int(t.dst_s.min()) >= 0 and int(t.dst_s.max()) < 300
    assert int(t.perm_dst.min()) >= 0 and int(t.perm_dst.max()) < bad.shape[1]
    assert int(t.rowptr_dst[-1]) == bad.shape[1] - 1 and int(t.rowptr_src[-1]) == bad.shape[1] - 2
    with pytest.raises(IndexError):
        t.resolve()
    net = gp.EncodeProcessDecode(2, 11, 3, 2, hidden_size=128).to(dev)
    g = gp.Graph(x=torch.randn(300, 11, device=dev), edge_attr=torch.randn(bad.shape[1], 3, device=dev), edge_index=bad.to(dev))
    with pytest.raises(IndexError):
        net(g)
    torch.cuda.synchronize()  # nothing faulted on the way
    g.edge_index = ei.to(dev)
    assert bool(torch.isfinite(net(g)).all())


def test_hub_graph_first_pass_is_optimistic_and_correct(dev):
    """a node with 5000 in-edges: the first pass over a lazily built topology takes the hub-free launches (a long
    segment is summed serially: slow, correct), later passes the chunked ones; both equal the oracle"""
    H, N, L = 128, 400, 2
    rng = np.random.default_rng(2)
    dst = np.concatenate([rng.integers(0, N, 3000), np.full(5000, 7)])
    src = rng.integers(0, N, dst.size)
    ei = torch.from_numpy(np.stack([src, dst]))
    params = R.make_params(R.epd_param_shapes(L, H, 11, 3, 2), 9)
    x_in, e_in = R.randn((N, 11), 1), R.randn((ei.shape[1], 3), 2)
    ref = O.epd_forward(x_in, e_in, ei, params, L)
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=H).to(dev)
    net.load_state_dict(params)
    g = gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=ei.to(dev))
    with torch.no_grad():
        first = net(g)           # fresh lazy topology through get_topology: optimistic pass
        topo = ops.get_topology(g.edge_index, N)
        assert topo.resolved and topo.has_hubs and topo.max_in_degree >= 5000
        second = net(g)          # resolved: chunked segment sums
    assert_close3(first.cpu(), ref, FWD_TOL, "hub graph, optimistic pass")
    assert_close3(second.cpu(), ref, FWD_TOL, "hub graph, resolved pass")


# ------------------------------------------------------------------ renumbering
@pytest.mark.parametrize("with_pos", [True, False])
def test_renumbered_forward_and_gradients_equal_the_oracle(dev, with_pos):
    L, H, N = 3, 128, 3000
    pos, ei, _ = R.delaunay_graph(N, 13)
    shuffle = torch.from_numpy(np.random.default_rng(0).permutation(N))     # numbering without locality
    inv = torch.empty_like(shuffle)
    inv[shuffle] = torch.arange(N)
    pos, ei = pos[shuffle], inv[ei]
    params = R.make_params(R.epd_param_shapes(L, H, 11, 3, 2), 4)
    x_in, e_in, cot = R.randn((N, 11), 1), R.randn((ei.shape[1], 3), 2), R.randn((N, 2), 3)
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.epd_forward(x_in, e_in, ei, P, L, act="silu")
    (ref * cot).sum().backward()
    old = ops.get_node_renumbering()
    outs, grads = {}, {}
    try:
        for mode in ("off", "on"):
            ops.set_node_renumbering(mode)
            with silu_models():
                net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=H).to(dev)
            net.load_state_dict(params)
            g = gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=ei.clone().to(dev), pos=pos.to(dev) if with_pos else None)
            out = net(g)
            (out * cot.to(dev)).sum().backward()
            topo = ops.get_topology(g.edge_index, N, pos=g.pos, renumber=True)
            assert (topo.node_order is not None) == (mode == "on")
            outs[mode], grads[mode] = out.detach().cpu(), {k: v.grad.cpu() for k, v in net.named_parameters()}
    finally:
        ops.set_node_renumbering(old)
    for mode in ("off", "on"):
        assert_close3(outs[mode], ref.detach(), FWD_TOL, f"renumbering {mode}")
        for k, gr in grads[mode].items():
            assert rel_err(gr, P[k].grad) < GRAD_TOL, (mode, k)
    assert rel_err(outs["on"], outs["off"]) < 2e-6


# ------------------------------------------------------------------ configs[3] as stated
@functools.lru_cache(maxsize=1)
def _c4_mesh():
    g = gp.square_mesh(1_000_000, seed=0)   # uniform points in generator order: the stated workload
    return g


def _closure(ei, N, seeds, L):
    """sub-mesh induced by the L-hop in-closure of ``seeds``: (nodes, kept edge ids, global -> local, sub edge_index);
    edge order preserved = the oracle sums in the same order"""
    src, dst = ei[0].numpy(), ei[1].numpy()
    inR = np.zeros(N, dtype=bool)
    inR[seeds] = True
    need_dst = inR.copy()
    for hop in range(L):
        if hop == L - 1:
            need_dst = inR.copy()
        inR[src[inR[dst]]] = True
    keep = need_dst[dst]
    nodes = np.nonzero(inR)[0]
    loc = np.full(N, -1, dtype=np.int64)
    loc[nodes] = np.arange(nodes.size)
    sub_ei = torch.from_numpy(np.stack([loc[src[keep]], loc[dst[keep]]]))
    assert int(sub_ei.min()) >= 0
    return nodes, np.nonzero(keep)[0], loc, sub_ei


def test_c4_unsorted_mesh_forward_with_engine_renumbering(dev):
    g = _c4_mesh()
    N, ei = g.x.shape[0], g.edge_index
    # the generator's numbering has no locality: neighbours are ~N/3 ids apart on average
    assert float((ei[0] - ei[1]).abs().double().mean()) > 0.2 * N
    L = 2
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 5)
    x_in = torch.randn(N, 11, generator=torch.Generator().manual_seed(1))
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
    net.load_state_dict(params)
    graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=ei.to(dev), pos=g.pos.to(dev))
    assert ops.get_node_renumbering() == "auto"
    with torch.no_grad():
        out = net(graph)                      # auto: 1M nodes >= RENUMBER_MIN_NODES -> Morton renumbering
        topo = ops.get_topology(graph.edge_index, N, pos=graph.pos, renumber=True)
        assert topo.node_order is not None and topo.resolved
        # locality after renumbering: the gathered source row of a dst-sorted edge row is near the row of its destination
        assert float((topo.src_s.long() - topo.dst_s.long()).abs().double().mean()) < 0.01 * N
        again = net(graph)
        assert torch.equal(out, again)        # deterministic
        graph2 = gp.Graph(x=graph.x, edge_attr=graph.edge_attr, edge_index=graph.edge_index)
        graph2.mgn_topology = ops.Topology(graph.edge_index, N)      # raw numbering
        raw = net(graph2)
    assert rel_err(out, raw) < 2e-6
    seeds = np.concatenate([np.arange(0, 300), np.arange(N // 2, N // 2 + 300), np.arange(N - 300, N)])
    nodes, kept, loc, sub_ei = _closure(ei, N, seeds, L)
    assert nodes.size < 40000
    ref = O.epd_forward(x_in[nodes], g.edge_attr[torch.from_numpy(kept)], sub_ei, params, L)
    assert_close3(out.cpu()[seeds], ref[loc[seeds]], FWD_TOL, "1M-node unsorted mesh, 2 rounds, seeds")


@pytest.mark.parametrize("recompute", ["off", "on"])
def test_c4_gradients_at_full_size_via_closure(dev, recompute):
    """every parameter gradient of the full 1M-node / 6M-edge step (rows past 2^31 bytes, k_wgrad_x6 and k_segsum2 at
    that size, the recompute path) against the oracle: the cotangent lives on 900 seed nodes, so the gradient is
    a function of their L-hop closure only"""
    g = _c4_mesh()
    N, ei = g.x.shape[0], g.edge_index
    L = 2
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 6)
    x_in = torch.randn(N, 11, generator=torch.Generator().manual_seed(2))
    seeds = np.concatenate([np.arange(0, 300), np.arange(N // 2, N // 2 + 300), np.arange(N - 300, N)])
    cot = R.randn((seeds.size, 2), 8)
    nodes, kept, loc, sub_ei = _closure(ei, N, seeds, L)
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.epd_forward(x_in[nodes], g.edge_attr[torch.from_numpy(kept)], sub_ei, P, L, act="silu")
    (ref[loc[seeds]] * cot).sum().backward()
    with silu_models():
        net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
    net.load_state_dict(params)
    graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=ei.to(dev), pos=g.pos.to(dev))
    old = ops.get_activation_recompute()
    try:
        ops.set_activation_recompute(recompute)
        out = net(graph)
        (out[torch.from_numpy(seeds).to(dev)] * cot.to(dev)).sum().backward()
    finally:
        ops.set_activation_recompute(old)
    assert_close3(out.detach().cpu()[seeds], ref.detach()[loc[seeds]], FWD_TOL, "forward on the seeds")
    worst = 0.0
    for k, p in net.named_parameters():
        e = rel_err(p.grad, P[k].grad)
        worst = max(worst, e)
        assert e < GRAD_TOL, (k, e)
    print(f"C4 gradients (recompute {recompute}): worst parameter {worst:.2e}")


def test_c4_gradients_15_rounds_via_closure(dev):
    """[r5] the config's own depth: every parameter gradient of the 15-round 1M-node training step -- partial activation recompute as
    "auto" picks it (5 of 15 rounds re-run), k_wgrad_pc and k_segsum2 at 6M rows -- against the oracle on the 15-hop closures of six
    seed nodes (SiLU network: no ReLU branch flips to blur the bar)"""
    g = _c4_mesh()
    N, ei = g.x.shape[0], g.edge_index
    L = 15
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 26)
    x_in = torch.randn(N, 11, generator=torch.Generator().manual_seed(22))
    seeds = np.array([0, 1, N // 2, N // 2 + 1, N - 2, N - 1])
    cot = R.randn((seeds.size, 2), 28)
    nodes, kept, loc, sub_ei = _closure(ei, N, seeds, L)
    assert nodes.size < 20000, nodes.size
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.epd_forward(x_in[nodes], g.edge_attr[torch.from_numpy(kept)], sub_ei, P, L, act="silu")
    (ref[loc[seeds]] * cot).sum().backward()
    with silu_models():
        net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
    net.load_state_dict(params)
    graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=ei.to(dev), pos=g.pos.to(dev))
    out = net(graph)
    (out[torch.from_numpy(seeds).to(dev)] * cot.to(dev)).sum().backward()
    assert_close3(out.detach().cpu()[seeds], ref.detach()[loc[seeds]], FWD_TOL, "forward on the seeds, 15 rounds")
    worst = 0.0
    for k, p in net.named_parameters():
        e = rel_err(p.grad, P[k].grad)
        worst = max(worst, e)
        assert e < GRAD_TOL, (k, e)
    print(f"C4 gradients, 15 rounds: worst parameter {worst:.2e}")


def test_c4_relu_gradients_at_full_size_via_closure(dev):
    """The DEFAULT activation at the 1M-node size (every shipped JSON is ReLU, the large-M gradient
    checks ran SiLU networks): same closure construction, ReLU network, flip-aware bar -- a gradient either agrees with the
    fp32 oracle to GRAD_TOL, or (a pre-activation within rounding of zero took the other branch) is as close to the fp64
    oracle as the fp32 oracle itself is (x2)."""
    g = _c4_mesh()
    N, ei = g.x.shape[0], g.edge_index
    L = 2
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 16)
    x_in = torch.randn(N, 11, generator=torch.Generator().manual_seed(12))
    seeds = np.concatenate([np.arange(0, 300), np.arange(N // 2, N // 2 + 300), np.arange(N - 300, N)])
    cot = R.randn((seeds.size, 2), 18)
    nodes, kept, loc, sub_ei = _closure(ei, N, seeds, L)

    def oracle(dtype):
        P = {k: v.clone().to(dtype).requires_grad_(True) for k, v in params.items()}
        ref = O.epd_forward(x_in[nodes].to(dtype), g.edge_attr[torch.from_numpy(kept)].to(dtype), sub_ei, P, L)
        (ref[loc[seeds]] * cot.to(dtype)).sum().backward()
        return ref.detach(), P

    ref32, P32 = oracle(torch.float32)
    _, P64 = oracle(torch.float64)
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)   # ReLU (the default)
    net.load_state_dict(params)
    graph = gp.Graph(x=x_in.to(dev), edge_attr=g.edge_attr.to(dev), edge_index=ei.to(dev), pos=g.pos.to(dev))
    out = net(graph)
    (out[torch.from_numpy(seeds).to(dev)] * cot.to(dev)).sum().backward()
    assert_close3(out.detach().cpu()[seeds], ref32[loc[seeds]], FWD_TOL, "forward on the seeds (ReLU)")
    worst32, flipped = 0.0, []
    for k, p in net.named_parameters():
        e32 = rel_err(p.grad, P32[k].grad)
        worst32 = max(worst32, e32)
        if e32 >= GRAD_TOL:
            h64, c64 = rel_err(p.grad, P64[k].grad), rel_err(P32[k].grad, P64[k].grad)
            assert h64 < 2.0 * c64 + 1e-6, (k, e32, h64, c64)
            flipped.append(k)
    print(f"C4 ReLU gradients: worst parameter vs the fp32 oracle {worst32:.2e}; {len(flipped)} parameter(s) judged against fp64")


# ------------------------------------------------------------------ noise with overlapping ranges
def test_noise_overlapping_ranges_apply_one_after_the_other(dev):
    from graph_physics_amd import preprocess as PP

    n = 5000
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, 7)).astype(np.float32)
    x[:, 6] = rng.choice([0.0, 0.0, 0.5, 4.0, 6.0], size=n)          # 0.5 is NOT NORMAL (the float is compared)
    starts, ends, scales = [0, 2, 1], [4, 5, 3], [0.1, 0.2, 0.05]     # overlapping column ranges
    want, _ = O.add_noise_oracle(x, starts, ends, scales, 6, seed=11, offset=3)
    g = gp.Graph(x=torch.from_numpy(x.copy()).to(dev))
    PP.add_noise(g, starts, ends, scales, 6, seed=11, offset=3)
    got = g.x.cpu().numpy()
    assert np.abs(got - want).max() < 2e-6
    assert np.array_equal(got[x[:, 6] != 0.0], x[x[:, 6] != 0.0])     # untouched rows, bit for bit
    with pytest.raises(RuntimeError, match="node_type_index"):
        PP.add_noise(g, [5], [7], [0.1], 6)                            # the type column inside a noised range


# ------------------------------------------------------------------ fused edge backward (opt-in kernel)
@pytest.mark.parametrize("E", [1, 31, 64, 97, 5000, 70001])
def test_fused_edge_backward_equals_the_split_launches(dev, E, monkeypatch):
    """mgn_edge_bwd_fused (backward chain + the four E-row weight gradients in one kernel) against the launches it
    replaces (mgn_mlp_bwd edge chain + mgn_wgrad), which the oracle tests pin: dZ0 / dE bit-identical (same chain
    arithmetic), weight / bias / scale gradients to fp32 summation order; ragged row counts (M mod 64 in {1, 31, 33},
    fewer rows than a tile, fewer tiles than workgroups).  (MGN_PPR=0: the x6 chain on the split side -- the register-resident-weights
    chain that takes launches from 65 536 rows agrees with it to rounding only, tests/test_hip_ppr.py.)"""
    from graph_physics_amd import _capi

    monkeypatch.setenv("MGN_PPR", "0")
    H, N = 128, max(8, E // 6)
    f = dict(dtype=torch.float32, device=dev)
    gen = torch.Generator(device="cpu").manual_seed(E)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(dev)  # noqa: E731
    dst = torch.sort(torch.randint(0, N, (E,), generator=gen)).values.to(torch.int32).to(dev)
    src = torch.randint(0, N, (E,), generator=gen).to(torch.int32).to(dev)
    x, e = rnd(N, H), rnd(E, H)
    W0, Wh = rnd(H, 3 * H) * 0.05, [rnd(H, H) * 0.09 for _ in range(3)]
    bs, sc = [rnd(H) * 0.1 for _ in range(4)], torch.rand(H, generator=gen).to(dev) + 0.5
    Pd, Ps = x @ W0[:, H:2 * H].t(), x @ W0[:, 2 * H:].t()
    pk = torch.empty(8 * _capi.WPACK_BYTES, dtype=torch.uint8, device=dev)
    u = [pk.data_ptr() + i * _capi.WPACK_BYTES for i in range(8)]
    ops.wpack([(W0.data_ptr(), 3 * H, False, u[0])] + [(Wh[l].data_ptr(), H, False, u[l + 1]) for l in range(3)] +
              [(Wh[2].data_ptr(), H, True, u[4]), (Wh[1].data_ptr(), H, True, u[5]), (Wh[0].data_ptr(), H, True, u[6]),
               (W0.data_ptr(), 3 * H, True, u[7])], dev)
    m, e_new = torch.empty(E, H, **f), torch.empty(E, H, **f)
    He, Ue, Re = [torch.empty(E, H, **f) for _ in range(3)], torch.empty(E, H, **f), torch.empty(E, **f)
    Me = [torch.empty(E, 4, dtype=torch.int32, device=dev) for _ in range(3)]
    ops.mlp_fwd(E, H, [(e, None, H)], [W0] + Wh, bs, sc, H, e, e_new, m, He, Ue, Re, ldw0=3 * H, adds=[(Pd, dst), (Ps, src)],
                wpk=u[:4], saveM=Me)
    de, dagg = rnd(E, H), rnd(N, H)
    nb = H // 16
    # split launches
    dZ = [torch.empty(E, H, **f) for _ in range(4)]
    de_s, dsc_s = torch.empty(E, H, **f), torch.empty(H, **f)
    gW_s = [torch.zeros(H, 3 * H, **f)] + [torch.empty(H, H, **f) for _ in range(3)]
    gb_s = [torch.empty(H, **f) for _ in range(4)]
    ops.mlp_bwd(E, H, 4, de, dagg, dst, H, Ue, Re, sc, He, [None] * 4, dZ, [(None, de, de_s)], [None] * 4, dsc_s, wpk=u[4:], Ms=Me)
    ops.wgrad([(dZ[0], H, nb, e, H, nb, H, gW_s[0], 0, 3 * H, gb_s[0])] +
              [(dZ[l], H, nb, He[l - 1], H, nb, H, gW_s[l], 0, H, gb_s[l]) for l in range(1, 4)], dev)
    # fused kernel
    dZ0, de_f, dsc_f = torch.full((E, H), float("nan"), **f), torch.full((E, H), float("nan"), **f), torch.empty(H, **f)
    gW_f = [torch.zeros(H, 3 * H, **f)] + [torch.empty(H, H, **f) for _ in range(3)]
    gb_f = [torch.empty(H, **f) for _ in range(4)]
    ops.edge_bwd_fused(E, de, dagg, dst, Ue, Re, sc, [e] + He, Me, u[4:], de_f, dZ0,
                       [(gW_f[0], 0, 3 * H), (gW_f[1], 0, H), (gW_f[2], 0, H), (gW_f[3], 0, H)], gb_f, dsc_f)
    assert torch.equal(dZ0, dZ[0]) and torch.equal(de_f, de_s)
    assert rel_err(dsc_f, dsc_s) < 2e-6
    for l in range(4):
        assert rel_err(gW_f[l][:, :H], gW_s[l][:, :H]) < 3e-6, l
        assert rel_err(gb_f[l], gb_s[l]) < 3e-6, l
    assert float(gW_f[0][:, H:].abs().max()) == 0.0      # only the first slab of the [128, 384] first-layer gradient is written


def test_fused_edge_backward_inside_the_training_step(dev):
    """MGN_FUSED_BWD=1 through the whole processor backward (read per call): gradients equal the default path's"""
    import os

    L, N = 3, 2500
    pos, ei, _ = R.delaunay_graph(N, 17)
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 3)
    x_in, e_in, cot = R.randn((N, 11), 1), R.randn((ei.shape[1], 3), 2), R.randn((N, 2), 3)
    grads = {}
    for flag in ("0", "1"):
        os.environ["MGN_FUSED_BWD"] = flag
        try:
            net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
            net.load_state_dict(params)
            g = gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=ei.to(dev))
            (net(g) * cot.to(dev)).sum().backward()
            grads[flag] = {k: p.grad.clone() for k, p in net.named_parameters()}
        finally:
            os.environ.pop("MGN_FUSED_BWD", None)
    for k in grads["0"]:
        assert rel_err(grads["1"][k], grads["0"][k]) < 5e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("M,nf,kw,lda,ldb", [(1, 64, 64, 64, 64), (17, 48, 20, 192, 64), (1000, 64, 64, 192, 128), (33333, 16, 4, 16, 4),
                                             (70001, 64, 60, 64, 64), (5, 32, 64, 68, 72)])
def test_row_vector_weight_gradient_kernel(dev, monkeypatch, M, nf, kw, lda, ldb):
    """k_wgrad_row64 (jobs up to 64 x 64 with 16-byte addressable rows: strided slabs of wider matrices, widths that are multiples
    of 4 but not of 16, ragged row counts, the bias by-product) against fp64 torch -- and bit-for-bit the same call run to run; the
    generic kernel (MGN_WGRAD_NO_ROW64) on the same jobs as a cross-check"""
    g = torch.Generator().manual_seed(M + nf + kw)
    A_full = torch.randn(M, lda, generator=g).to(dev)
    B_full = torch.randn(M, ldb, generator=g).to(dev)
    A, B = A_full[:, lda - nf:], B_full[:, :kw]          # a slab at a column offset (16-byte aligned: lda - nf is a multiple of 4)
    nja, nkb = (nf + 15) // 16, (kw + 15) // 16
    ref = (A.double().t() @ B.double())
    ref_b = A.double().sum(0)

    def run():
        dW = torch.full((16 * nja, 16 * nkb), float("nan"), device=dev)
        db = torch.full((16 * nja,), float("nan"), device=dev)
        ops.wgrad([(A, lda, nja, B, ldb, nkb, kw, dW, 0, 16 * nkb, db)], dev)
        return dW, db
    dW, db = run()
    dW2, db2 = run()
    assert torch.equal(dW, dW2) and torch.equal(db, db2)
    scale = float(ref.abs().max()) + 1e-30
    assert float((dW[:nf, :kw].double() - ref).abs().max()) / scale < 2e-6
    assert float((db[:nf].double() - ref_b).abs().max()) / (float(ref_b.abs().max()) + 1e-30) < 2e-6
    assert float(dW[:nf, kw:].abs().max()) == 0.0 if kw < 16 * nkb else True   # columns past kw are zeros, not garbage
    monkeypatch.setenv("MGN_WGRAD_NO_ROW64", "1")
    dW3, db3 = run()
    assert float((dW3[:nf, :kw] - dW[:nf, :kw]).abs().max()) / scale < 2e-6


@pytest.mark.gpu
def test_optimizer_table_form_equals_the_chunked_form(dev, monkeypatch):
    """mgn_clip_adamw_t (two launches, parameter table on the device, gradient pointers as kernel arguments) against mgn_clip_adamw
    (96 tensors per launch): parameters, both moments, clipped gradients and the norm bit for bit over three steps with fresh
    gradient tensors each step, 376 tensors as the 15-round model has them (sizes 1 .. 49 152)"""
    from graph_physics_amd import harness

    shapes = [(128, 384), (128,), (128, 128), (128,), (128, 128), (128,), (128, 128), (128,), (128,), (128, 256), (1,), (2, 128)] * 31 + [(7, 3)] * 4
    res = {}
    for form in ("table", "chunked"):
        if form == "chunked":
            monkeypatch.setenv("MGN_OPT_NO_TABLE", "1")
        else:
            monkeypatch.delenv("MGN_OPT_NO_TABLE", raising=False)
        gen = torch.Generator().manual_seed(3)
        params = [torch.nn.Parameter(torch.randn(*s, generator=gen).to(dev)) for s in shapes]
        opt = harness.FusedClipAdamW(params, 1e-3, max_norm=1.0)
        for step in range(3):
            for p in params:
                p.grad = (torch.randn(*p.shape, generator=gen) * (10.0 if step == 1 else 0.01)).to(dev)   # step 1 clips, the others do not
            opt.step()
        assert (opt._table is not None) == (form == "table")
        res[form] = ([p.detach().clone() for p in params], [m.clone() for m in opt.exp_avg], [v.clone() for v in opt.exp_avg_sq],
                     [p.grad.clone() for p in params], opt.norm_t.clone())
    monkeypatch.delenv("MGN_OPT_NO_TABLE", raising=False)
    for a, b in zip(res["table"][:4], res["chunked"][:4]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(res["table"][4], res["chunked"][4])


@pytest.mark.gpu
@pytest.mark.parametrize("N,O", [(30160, 2), (1, 3), (70001, 3), (257, 1)])
def test_fused_masked_mse_equals_the_torch_formula(dev, monkeypatch, N, O):
    """harness.l2_loss on the device (mgn_masked_mse_fwd / _bwd: three launches) against its own torch formula (MGN_TORCH_LOSS) and the
    oracle: loss and gradient, node types read through a strided view as the training step passes them, an upstream factor in the
    backward pass; a batch without a single selected row gives nan like torch's mean of an empty selection"""
    from graph_physics_amd import harness

    g = torch.Generator().manual_seed(N + O)
    x = torch.randn(N, 5, generator=g)
    x[:, 2] = torch.randint(0, 7, (N,), generator=g).float()
    x = x.to(dev)
    nt = x[:, 2]                                           # stride 5
    out = torch.randn(N, O, generator=g).to(dev).requires_grad_(True)
    tgt = torch.randn(N, O, generator=g).to(dev)
    res = {}
    for form in ("fused", "torch"):
        if form == "torch":
            monkeypatch.setenv("MGN_TORCH_LOSS", "1")
        else:
            monkeypatch.delenv("MGN_TORCH_LOSS", raising=False)
        out.grad = None
        loss = harness.l2_loss(out, tgt, nt)
        (3.0 * loss).backward()
        res[form] = (loss.detach().clone(), out.grad.clone())
    monkeypatch.delenv("MGN_TORCH_LOSS", raising=False)
    ref = O_l2(out.detach().cpu(), tgt.cpu(), nt.cpu())
    if bool(torch.isnan(ref)):
        assert bool(torch.isnan(res["fused"][0]))
        return
    assert abs(float(res["fused"][0]) - float(ref)) < 2e-6 * abs(float(ref))
    assert abs(float(res["fused"][0]) - float(res["torch"][0])) < 2e-6 * abs(float(ref))
    assert rel_err(res["fused"][1], res["torch"][1]) < 1e-6


def O_l2(out, tgt, nt):
    return O.l2_loss(out, tgt, nt)
