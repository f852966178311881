"""CPU: host-side mirror of the reference interface -- constructor signatures,
state_dict keys, error conventions, JSON factory, harness formulas -- and the rule
that the product path fails loudly instead of falling back to the CPU."""
import pytest
import torch

import recipe as R
import graph_physics_amd as gp
from graph_physics_amd import harness
from oracle import mgn_oracle as O


def test_state_dict_keys_match_reference_layout():
    net = gp.EncodeProcessDecode(15, 11, 3, 2, hidden_size=128)
    sd = net.state_dict()
    want = R.epd_param_shapes(15, 128, 11, 3, 2)
    assert list(sd.keys()) == list(want.keys())
    for k, s in want.items():
        assert tuple(sd[k].shape) == s
    assert sum(v.numel() for v in sd.values()) == 2873730
    assert sd["processor_list.0.edge_block.0.weight"].shape == (128, 384)
    assert sd["processor_list.0.node_block.0.weight"].shape == (128, 256)
    assert "decode_module.7.scale" not in sd
    assert isinstance(net.processor_list, torch.nn.ModuleList) and net.hidden_size == 128 and net.d == 2
    for attr in ("nodes_encoder", "edges_encoder", "decode_module"):
        assert isinstance(getattr(net, attr), torch.nn.Module)
    net.load_state_dict(R.make_params(want, 0))


def test_only_processor_has_no_encoders():
    net = gp.EncodeProcessDecode(2, 16, 16, 16, hidden_size=16, only_processor=True)
    assert not hasattr(net, "nodes_encoder")
    assert len(net.processor_list) == 2


def test_error_conventions():
    with pytest.raises(AssertionError):
        gp.build_mlp(4, 8, 4, nb_of_layers=1)  # layers.py:188
    with pytest.raises(ValueError):
        gp.GraphNetBlock(16, use_rope=True, rope_axes=5)  # layers.py:965
    with pytest.raises(ValueError):
        gp.EncodeProcessDecode(1, 2, 2, 2, use_rope_embeddings=True, rope_pos_dimension=4)  # processors.py:113
    with pytest.raises(ValueError, match="Model type 'foo' not supported."):
        gp.get_model({"model": {"type": "foo", "node_input_size": 2}})  # parse_parameters.py:162
    with pytest.raises(ValueError, match="too small"):
        gp.GraphNetBlock(4, use_rope=True, rope_axes=3)  # layers.py:968-971
    with pytest.raises(NotImplementedError, match="transolver"):
        gp.get_model({"model": {"type": "transolver", "node_input_size": 2}})  # outside the message-passing path
    with pytest.raises(AssertionError):
        gp.Attention(64, 64, num_heads=5)  # layers.py:600-602
    blk = gp.GraphNetBlock(16, use_rope=True)
    with pytest.raises(ValueError, match="pos"):
        blk(torch.zeros(3, 16), torch.zeros(2, 2, dtype=torch.int64), torch.zeros(2, 16))  # layers.py:1021-1024


def test_block_variants_construct_with_reference_state_dict_keys():
    """N3: the JSON keys use_silu_activation / use_gated_attention / use_gated_mlp / use_rope_embeddings
    construct, with the reference's parameter names (layers.py:932-987, 213-278)."""
    b = gp.GraphNetBlock(32, use_gate=True, use_rope=True, rope_axes=2)
    keys = set(b.state_dict())
    assert {"gate_pos", "gate_proj.weight", "gate_proj.bias"} <= keys and "_rope_inv_freq" not in keys
    assert b._rope_inv_freq.numel() == 32 // 4 and abs(float(b._rope_inv_freq[1]) - 10000.0 ** (-1 / 8)) < 1e-7
    g = gp.GraphNetBlock(32, use_gated_mlp=True)
    with pytest.raises(NotImplementedError, match="hidden_size in"):     # widths off the dense kernels: said at construction
        gp.GraphNetBlock(256, use_gated_mlp=True)
    assert {"edge_block.0.scale", "edge_block.1.linear1.weight", "edge_block.1.linear2.bias", "edge_block.2.weight"} <= set(g.state_dict())
    assert g.edge_block[1].linear1.weight.shape == (96, 96) and g.node_block[2].weight.shape == (32, 96)
    b3 = gp.GraphNetBlock(32, nb_of_layers=3, layer_norm=False)
    assert set(b3.edge_block.state_dict()) == {"0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias"}
    cfg = gp.cylinder_config(2, 32)
    cfg["model"].update(use_silu_activation=True, use_gated_attention=True, use_rope_embeddings=True, rope_pos_dimension=2)
    try:
        net = gp.get_model(cfg)
        assert net.processor_list[0].spec.act == "silu" and net.nodes_encoder.act == "silu"
        assert net.processor_list[0].use_gate and net.processor_list[0].use_rope
    finally:
        gp.layers.set_use_silu_activation(False)
    # N4: the Transformer family through the same factory, reference parameter names
    t = gp.get_model({"model": {"type": "transformer", "message_passing_num": 2, "hidden_size": 64, "node_input_size": 14, "output_size": 3,
                                "edge_input_size": 0, "num_heads": 4, "use_rope_embeddings": True, "use_gated_attention": True},
                      "training": {"use_temporal_block": True}})   # training_config/coarse-aneurysm.json shape
    keys = set(t.state_dict())
    assert {"processor_list.0.attention.q_proj.weight", "processor_list.0.attention.rope_inv_freq", "processor_list.0.attention.gate_proj.bias",
            "processor_list.1.norm2.scale", "processor_list.0.gated_mlp.1.linear2.weight", "temporal_block.mixer.2.bias",
            "temporal_block.gate.0.weight", "decode_module.6.bias"} <= keys
    assert not any(k.startswith("edges_encoder") for k in keys) and t.nodes_encoder[0].in_features == 14 + 9
    assert gp.EncodeProcessDecode(2, 11, 3, 2, hidden_size=32, use_temporal_block=True).temporal_block.H == 4


def test_json_factory():
    cfg = gp.cylinder_config()
    net = gp.get_model(cfg)
    assert net.nodes_encoder[0].in_features == 2 + 9  # NodeType.SIZE added, parse_parameters.py:96
    assert len(net.processor_list) == 15 and net.hidden_size == 128
    cfg["model"].update(message_passing_num=5, hidden_size=32)  # the shipped cylinder.json values
    assert gp.get_model(cfg).hidden_size == 32


def test_cpu_tensors_fail_loudly():
    net = gp.EncodeProcessDecode(1, 11, 3, 2, hidden_size=16)
    _, ei, ea = R.delaunay_graph(12, 0)
    g = gp.Graph(x=torch.randn(12, 11), edge_attr=ea, edge_index=ei)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(g)
    blk = gp.GraphNetBlock(16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blk(torch.randn(12, 16), ei, torch.randn(ei.shape[1], 16))


def test_product_never_imports_oracle():
    import os
    import re

    root = os.path.dirname(gp.__path__[0]) + "/graph-physics_amd"
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
    bench = open(os.path.join(os.path.dirname(root), "bench.py")).read() if os.path.exists(os.path.join(os.path.dirname(root), "bench.py")) else ""
    for line in bench.splitlines():
        if re.match(r"\s*(from|import)\s+oracle", line):
            assert "cpu_baseline" in bench


def test_harness_formulas_match_oracle():
    for step in (0, 1, 3, 4, 50, 99, 150):
        assert harness.lr_factor(step, 4, 100) == O.lr_factor(step, 4, 100)
    torch.manual_seed(0)
    out, tgt = torch.randn(50, 2), torch.randn(50, 2)
    nt = torch.randint(0, 7, (50,)).float()
    assert torch.allclose(harness.l2_loss(out, tgt, nt), O.l2_loss(out, tgt, nt), rtol=1e-6)
    m = harness.build_mask(nt)
    assert torch.equal(m, ~((nt == 0) | (nt == 5)))


def test_normalizer_matches_oracle_state():
    n = gp.Normalizer(3, device="cpu")
    o = O.NormalizerState(3)
    torch.manual_seed(1)
    for _ in range(3):
        d = torch.randn(20, 3) * 2 + 1
        assert torch.allclose(n(d, True), o(d, True), rtol=1e-6, atol=1e-7)
    d = torch.randn(5, 3)
    assert torch.allclose(n.inverse(n(d, False)), d, atol=1e-6)  # reference test_layers.py:92-100
    assert set(n.state_dict().keys()) == {"_acc_count", "_num_accumulations", "_acc_sum", "_acc_sum_squared"}


def test_mesh_contract():
    g = gp.cylinder_mesh(300, seed=3)
    ei = g.edge_index
    key = ei[0] * 300 + ei[1]
    assert torch.all(key[1:] > key[:-1])  # coalesced: sorted by (src,dst), no duplicates
    assert torch.all(ei[0] != ei[1])
    rev = set(map(tuple, ei.t().tolist()))
    assert all((b, a) in rev for a, b in rev)  # symmetric closure
    assert g.edge_attr.shape == (ei.shape[1], 3) and g.x.shape == (300, 4)
    b = gp.collate([g, gp.cylinder_mesh(200, seed=4)])
    assert b.x.shape[0] == 500 and int(b.edge_index.max()) < 500 and int(b.edge_index[:, ei.shape[1]:].min()) >= 300


def test_bench_refuses_a_gpus_flag_that_contradicts_the_launch():
    """`--gpus N` must equal WORLD_SIZE under an external launcher (never silently ignored); checked before
    anything touches a GPU."""
    import os, subprocess, sys
    from conftest import REPO as REPO_ROOT
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO_ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


# ---------------------------------------------------------------- round 3: host logic of the new options
def test_partitioned_model_refuses_what_it_would_silently_drop():
    """[r6] a model with the temporal block needs the plan of the flipped edge list (attention rows = sources) built with the same
    partition vector -- without it, or with another partition, the constructor refuses instead of computing another function; RoPE
    constructs [r4] and asks for the owned positions like the reference asks for graph.pos."""
    from graph_physics_amd import distributed as D
    from graph_physics_amd import partition as P

    pos, ei, _ = R.delaunay_graph(60, 3)
    plan = P.build_rank_plan(ei, P.partition_nodes(pos.numpy(), ei, 2), 0, 2)
    part = P.partition_nodes(pos.numpy(), ei, 2)
    tnet = gp.EncodeProcessDecode(2, 11, 3, 2, hidden_size=32, use_temporal_block=True, attention_backend="dgl")
    with pytest.raises(ValueError, match="temporal_plan"):
        D.PartitionedEPD(tnet, plan)
    with pytest.raises(ValueError, match="same partition"):
        D.PartitionedEPD(tnet, plan, temporal_plan=P.build_rank_plan(ei.flip(0), 1 - part, 0, 2))
    D.PartitionedEPD(tnet, plan, temporal_plan=P.build_rank_plan(ei.flip(0), part, 0, 2))
    # an installation without DGL: the block gets no adjacency and is row-wise -- no second plan
    D.PartitionedEPD(gp.EncodeProcessDecode(2, 11, 3, 2, hidden_size=32, use_temporal_block=True, attention_backend="pyg"), plan)
    with pytest.raises(NotImplementedError, match="sparse-attention"):
        D.PartitionedETD(gp.EncodeTransformDecode(1, 11, 2, hidden_size=32, attention_backend="pyg"), plan)
    D.PartitionedEPD(gp.EncodeProcessDecode(2, 11, 3, 2, hidden_size=32, use_rope_embeddings=True, rope_pos_dimension=2), plan)
    pm = D.PartitionedEPD(gp.EncodeProcessDecode(2, 11, 3, 2, hidden_size=32, use_gated_attention=True), plan)
    with pytest.raises(ValueError, match="phi_own"):
        pm(torch.zeros(plan.n_own, 11), torch.zeros(plan.edge_ids.numel(), 3), phi_own=torch.zeros(plan.n_own + 1))


def test_node_renumbering_switch():
    from graph_physics_amd import ops

    old = ops.get_node_renumbering()
    try:
        with pytest.raises(ValueError):
            ops.set_node_renumbering("maybe")
        ops.set_node_renumbering("auto")
        assert not ops.want_renumbering(30160) and ops.want_renumbering(1_000_000)
        ops.set_node_renumbering("off")
        assert not ops.want_renumbering(1_000_000)
        ops.set_node_renumbering("on")
        assert ops.want_renumbering(10)
    finally:
        ops.set_node_renumbering(old)


def test_rank_plan_with_positions_keeps_the_contract():
    """build_rank_plan(pos=...) only changes the ORDER inside the interior / boundary groups (Morton curve instead of
    global id): same owned set, interior first, same ghosts, same exchange lists up to that order."""
    import numpy as np
    from graph_physics_amd import partition as P

    pos, ei, _ = R.delaunay_graph(400, 8)
    part = P.partition_nodes(pos.numpy(), ei, 4)
    k = P.morton_keys(pos.numpy())
    assert k.shape == (400,) and len(np.unique(k)) == 400
    for r in range(4):
        a = P.build_rank_plan(ei, part, r, 4)
        b = P.build_rank_plan(ei, part, r, 4, pos=pos.numpy())
        assert (a.n_own, a.n_ghost, a.n_interior, a.n_interior_edges) == (b.n_own, b.n_ghost, b.n_interior, b.n_interior_edges)
        assert sorted(a.owned[:a.n_interior].tolist()) == sorted(b.owned[:b.n_interior].tolist())
        assert sorted(a.owned[a.n_interior:].tolist()) == sorted(b.owned[b.n_interior:].tolist())
        assert torch.equal(a.ghost, b.ghost) and a.send_counts == b.send_counts and a.recv_counts == b.recv_counts
        assert torch.equal(a.edge_ids, b.edge_ids)
        # the owned rows a peer asks for are the same GLOBAL nodes in the same order
        assert torch.equal(a.owned[a.send_idx], b.owned[b.send_idx])
        own_sorted = np.sort(b.owned.numpy())                      # keys are taken on the rank's own bounding box
        kown = dict(zip(own_sorted.tolist(), P.morton_keys(pos.numpy()[own_sorted]).tolist()))
        kb = np.array([kown[i] for i in b.owned[:b.n_interior].tolist()], dtype=np.uint64)
        assert bool((kb[1:] >= kb[:-1]).all())


def test_overlapped_grad_all_reduce_refuses_double_reports_and_ignores_foreign_parameters(monkeypatch):
    """the bucket listener is process-global -- a parameter reported twice before the step's all-reduce must raise
    (the bucketed value would silently replace an accumulated gradient), gradients of another model must be ignored, and a
    lazily built topology that reported stray indices must raise on every use, also from the cache."""
    from graph_physics_amd import distributed as D, ops

    mine = [torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(3))]
    other = torch.nn.Parameter(torch.zeros(5))
    sync = D.OverlappedGradAllReduce(bucket_bytes=1 << 30, params=mine)
    try:
        monkeypatch.setattr(sync, "_active", lambda: True)
        sync._on_ready([(mine[0], torch.ones(4)), (other, torch.ones(5))])
        assert [ptr for ptr, _ in sync._cur] == [mine[0].data_ptr()]          # the foreign gradient never entered a bucket
        sync._on_ready([(mine[1], torch.ones(3))])
        with pytest.raises(RuntimeError, match="reported twice"):
            sync._on_ready([(mine[0], torch.ones(4))])
        sync.reset()                                                           # an aborted step: start clean
        assert sync._cur == [] and sync._seen == set() and sync._inflight == []
        sync._on_ready([(mine[0], torch.ones(4))])
    finally:
        sync.close()
    assert ops._grad_ready_hook is None if hasattr(ops, "_grad_ready_hook") else True


def test_recompute_env_is_validated():
    from graph_physics_amd import ops

    assert ops._parse_recompute_env("6") == 6 and ops._parse_recompute_env(" auto ") == "auto"
    with pytest.raises(ValueError):
        ops._parse_recompute_env("onn")


def test_round5_host_logic_recompute_size_rollout_modes_and_bench_ceilings():
    """host-side pieces of round 5 that need no GPU: the two-byte saves are priced in the recompute estimate (advisor r4), the
    rollout's replay switch validates its argument and falls back to the eager loop for frames it cannot key (CPU tensors,
    different edge_index tensors), the streaming ceiling of a roofline object interpolates the read : write probe, and the PMC
    traffic summary matches the scatter kernel's current name (`k_segsum2<8, false>` matched nothing)."""
    import importlib.util
    import os
    from graph_physics_amd import ops

    E, N, H, NL = 1000, 100, 128, 4
    full, half = ops.saved_activation_bytes(E, N, H, NL, 1, 0), ops.saved_activation_bytes(E, N, H, NL, 1, 0, save16=True)
    assert full - half == E * 2 * H * (NL - 1)          # H1..H3 of the EDGE rows at two bytes per value; node rows unchanged

    eng = harness.Engine.__new__(harness.Engine)       # no device needed for the key logic
    eng.grad_sync = None
    g = gp.cylinder_mesh(64, 0)
    frames = [g, g, g, g]
    assert eng._rollout_graph_key(frames) is None       # CPU tensors: never replayed
    with pytest.raises(ValueError):
        eng.rollout(frames, graph="maybe")
    with pytest.raises(ValueError):
        eng.rollout(frames, graph="on")                 # "on" demands frames it can key

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.rw_ceiling_gbps(0.0) == pytest.approx(6370.0) and bench.rw_ceiling_gbps(1.0) == pytest.approx(4516.0)
    assert 4890.0 <= bench.rw_ceiling_gbps(0.4) <= 4935.0 and bench.rw_ceiling_gbps(-1) == bench.rw_ceiling_gbps(0)
    o = bench.hbm_obj("k", 0.1, 1e9, traffic=None, write_bytes=0.0)
    assert o["read_bytes"] == int(1e9) and o["write_bytes"] == 0 and o["ceiling_gbps"] == 6370.0 and o["frac"] == pytest.approx(1e9 / 1e-4 / 1e9 / 8000.0, rel=1e-3)

    src = open(os.path.join(repo, "tools", "pmc_traffic_summary.py")).read()
    keys = [ln for ln in src.splitlines() if "k_segsum2<8" in ln and "KEYS" not in ln[:4]]
    assert any('("k_segsum2<8", None)' in ln for ln in src.splitlines())
    assert "k_segsum2<8, false>".startswith("k_segsum2<8") and not "k_segsum2<8, false>".startswith("k_segsum2<8>")
    assert '("k_wgrad_pc", 256)' in src


def test_transformer_node_renumbering_switch_and_binding_layout():
    """[r5] host side of the Transformer path's additions: the renumbering switch (ops.set_node_renumbering governs both model
    families; positions must be on the device; auto starts at transformer.ATTN_RENUMBER_MIN_NODES), and the ctypes mirrors of the two
    structs that grew this round end with the new fields in the header's order (a stale binding would shift every later argument)."""
    from graph_physics_amd import _capi, ops, transformer as T

    pos = torch.zeros(T.ATTN_RENUMBER_MIN_NODES, 3)
    prev = ops.get_node_renumbering()
    try:
        for mode in ("off", "on", "auto"):
            ops.set_node_renumbering(mode)
            assert T.want_attn_renumbering(pos.shape[0], pos) is False          # CPU positions: never
            assert T.want_attn_renumbering(pos.shape[0], None) is False
    finally:
        ops.set_node_renumbering(prev)
    names = [f[0] for f in _capi.LinearArgs._fields_]
    assert names[-10:] == ["precision", "w_transposed", "gb_z1", "gb_z2", "out2", "norm_scale_outer", "inv_outer_out", "z16", "x16", "out16"]
    assert [f[0] for f in _capi.RownormPhase._fields_] == ["x", "ldx", "K", "idx", "dx", "lddx", "acc"]
    import os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mgn_hip.h")).read()
    body = hdr[hdr.index("typedef struct {\n  int64_t M;\n  const float* x; int ldx; int K1;"):hdr.index("} mgn_linear_args;")]
    order = [body.index(k) for k in ("int precision;", "int w_transposed;", "const float* gb_z1;", "const float* gb_z2;", "float* out2;",
                                     "const float* norm_scale_outer;", "float* inv_outer_out;", "int z16;", "int x16;", "int out16;")]
    assert order == sorted(order)
    assert _capi.EXPECTED_VERSION == 136
