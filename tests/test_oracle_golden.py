"""CPU: the oracle (oracle/mgn_oracle.py) against the golden vectors minted from the
reference (tests/golden/make_golden.py).  Forward results are bit-exact; gradients
carry a tolerance because CPU index_put_(accumulate) is order-nondeterministic."""
import os

import numpy as np
import torch

import recipe as R
from conftest import load_golden, rel_err
from oracle import mgn_oracle as O


def _block_case(tag, H, N, seed):
    g = load_golden(tag)
    _, ei, _ = R.delaunay_graph(N, seed)
    assert torch.equal(ei, g["edge_index"])
    params = R.make_params(R.epd_param_shapes(1, H, 1, 1, 1, only_processor=True), seed)
    x = R.randn((N, H), seed + 1)
    e = R.randn((ei.shape[1], H), seed + 2)
    return g, ei, params, x, e


def test_block_forward_bit_exact():
    for tag, H, N, seed in (("block_h128", 128, 24, 11), ("block_h16", 16, 12, 12)):
        g, ei, params, x, e = _block_case(tag, H, N, seed)
        x2, e2, inter = O.graph_net_block(x, e, ei, params, "processor_list.0.", return_intermediates=True)
        assert torch.equal(x2, g["x_out"]) and torch.equal(e2, g["e_out"])
        assert torch.equal(inter["m"], g["m"]) and torch.equal(inter["agg"], g["agg"])


def test_block_backward():
    tag, H, N, seed = "block_h128", 128, 24, 11
    g, ei, params, x, e = _block_case(tag, H, N, seed)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    x.requires_grad_(True), e.requires_grad_(True)
    x2, e2 = O.graph_net_block(x, e, ei, p, "processor_list.0.")
    ((x2 * R.randn((N, H), seed + 3)).sum() + (e2 * R.randn((ei.shape[1], H), seed + 4)).sum()).backward()
    assert rel_err(x.grad, g["dx"]) < 1e-5 and rel_err(e.grad, g["de"]) < 1e-5
    for k, v in p.items():
        kk = "g_" + k[len("processor_list.0."):]
        if v.dim() == 2:
            assert rel_err(v.grad[:8], g[kk + "__rows8"]) < 1e-5
            assert abs(float(v.grad.norm()) - float(g[kk + "__norm"])) < 1e-5 * float(g[kk + "__norm"])
        else:
            assert rel_err(v.grad, g[kk]) < 1e-5


def test_epd_forward_bit_exact():
    for tag, L, N, seed in (("epd_l2", 2, 256, 21), ("epd_l15", 15, 256, 22)):
        g = load_golden(tag)
        _, ei, ea = R.delaunay_graph(N, seed)
        params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), seed)
        per = []
        out = O.epd_forward(R.randn((N, 11), seed + 1), R.randn((ea.shape[0], 3), seed + 2), ei, params, L, per_round=per)
        assert torch.equal(out, g["out"])
        assert torch.equal(torch.stack([p[0] for p in per]), g["x_round_row0"])


def test_edge_cases_bit_exact():
    H, L, N, E, seed = 128, 3, 40, 150, 31
    ei = R.random_graph(N, E, seed)
    g = load_golden("epd_random_graph")
    assert torch.equal(ei, g["edge_index"])
    params = R.make_params(R.epd_param_shapes(L, H, 11, 3, 2), seed)
    out = O.epd_forward(R.randn((N, 11), seed + 1), R.randn((E, 3), seed + 2), ei, params, L)
    assert torch.equal(out, g["out"])
    g = load_golden("epd_only_processor")
    params = R.make_params(R.epd_param_shapes(2, H, 1, 1, 1, only_processor=True), seed + 5)
    out = O.epd_forward(R.randn((N, H), seed + 6), R.randn((E, H), seed + 7), ei, params, 2, only_processor=True)
    assert torch.equal(out, g["out"])


def test_train_steps():
    H, L, N, seed = 128, 3, 96, 41
    g = load_golden("train_2steps")
    pos, ei, ea, xs, ys = R.trajectory(N, 3, seed)
    p = {k: v.clone().requires_grad_(True) for k, v in R.make_params(R.epd_param_shapes(L, H, 11, 3, 2), seed).items()}
    sim = O.SimulatorOracle(R.CYL_INDEX, 11, 3, 2)
    log = O.train_steps(p, sim, [(xs[t], ys[t], ea, ei) for t in range(2)], L, 1e-3, 4, 100)
    for t in range(2):
        assert abs(log[t][0] - float(g["loss"][t])) < 1e-5 * float(g["loss"][t])
        assert abs(log[t][1] - float(g["grad_norm"][t])) < 1e-5 * float(g["grad_norm"][t])
    assert torch.allclose(p["decode_module.6.weight"].detach(), g["w_last"], rtol=1e-5, atol=2e-6)
    assert torch.allclose(p["nodes_encoder.0.bias"].detach(), g["b_first"], rtol=1e-5, atol=2e-6)
    sums = np.array([p[k].detach().double().sum().item() for k in p])
    assert np.allclose(sums, g["param_sum"].numpy(), rtol=1e-6, atol=1e-4)
    assert torch.equal(sim.node_norm.acc_sum, g["node_norm_sum"])
    # learning-rate schedule (scheduler.py:51-67): lr_after[t] is the LR set for step t+1
    for t in range(2):
        assert abs(1e-3 * O.lr_factor(t + 1, 4, 100) - float(g["lr_after"][t])) < 1e-12


def test_rollout_bit_exact():
    H, L, N, seed, T = 128, 3, 96, 51, 5
    g = load_golden("rollout_5steps")
    pos, ei, ea, xs, ys = R.trajectory(N, T, seed)
    params = R.make_params(R.epd_param_shapes(L, H, 11, 3, 2), seed)
    sim = O.SimulatorOracle(R.CYL_INDEX, 11, 3, 2)
    nsd = {k[len("norm."):]: v for k, v in g.items() if k.startswith("norm.")}
    sim.out_norm.load(nsd, "_output_normalizer.")
    sim.node_norm.load(nsd, "_node_normalizer.")
    sim.edge_norm.load(nsd, "_edge_normalizer.")
    preds = O.rollout(params, sim, xs, ys, ea, ei, L)
    assert torch.equal(preds[0], g["pred1"]) and torch.equal(preds[1], g["pred2"]) and torch.equal(preds[4], g["pred5"])


def test_csr_oracle():
    key = torch.tensor([2, 0, 2, 1, 0, 2, 4])
    rowptr, perm = O.csr_by_key(key, 5)
    assert rowptr.tolist() == [0, 2, 3, 6, 6, 7]
    assert perm.tolist() == [1, 4, 3, 0, 2, 5, 6]


def test_faces_to_edges_oracle_on_reference_mesh():
    """The reference's own test mesh (tests/mock_vtu/cylinder_0.vtu, minted into
    golden/cylinder_vtu_mesh.npz by make_mesh_fixture.py): 1923 nodes, 3612 triangles ->
    11 070 directed edges (Euler count with one hole), symmetric, sorted by
    (src, dst), no duplicates, no self loops; and the host-side mesh builder agrees."""
    import numpy as np
    from graph_physics_amd import mesh

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cylinder_vtu_mesh.npz"))
    face, N = d["face"].astype(np.int64), d["pos"].shape[0]
    ei = O.faces_to_edges_oracle(face, N)
    assert N == 1923 and face.shape == (3, 3612) and ei.shape == (2, 11070)
    key = ei[0] * N + ei[1]
    assert (np.diff(key) > 0).all() and (ei[0] != ei[1]).all()
    assert np.array_equal(np.sort(ei[1] * N + ei[0]), key)  # symmetric closure
    assert np.array_equal(mesh.faces_to_edges(face.T, N), ei)
    ea = O.edge_features_oracle(torch.from_numpy(d["pos"]), torch.from_numpy(ei))
    assert ea.shape == (11070, 3) and torch.allclose(ea[:, 2], ea[:, :2].norm(dim=1))
    assert torch.equal(ea, mesh.edge_features(torch.from_numpy(d["pos"]), torch.from_numpy(ei)))


# ---------------------------------------------------------------- GraphNetBlock variants (N3)
import pytest  # noqa: E402


def variant_fixture():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "block_variants.npz"))
    return {k: (str(z[k]) if z[k].dtype.kind == "U" else torch.from_numpy(z[k])) for k in z.files}


@pytest.mark.parametrize("name", list(R.VARIANTS))
def test_block_variants_oracle_vs_golden(name):
    """the oracle's restatement of every GraphNetBlock variant (SiLU, gate, RoPE, gated MLP, 3 layers
    without norm, combination) against the fixtures minted from the reference (make_golden_variants.py):
    forward bit-exact, gradients 1e-5 (CPU index_put_ order)."""
    import graph_physics_amd as gp

    g = variant_fixture()
    v = R.VARIANTS[name]
    H, N, E, seed = 128, 40, 150, 300 + v["seed"]
    ei = R.random_graph(N, E, seed)
    keys = g[name + ".blk.keys"].split("|")
    # parameter shapes from a freshly built product module: its layout must be the reference's
    gp.layers.set_use_silu_activation(v["act"] == "silu")
    try:
        blk = gp.GraphNetBlock(H, **R.block_kwargs(v))
    finally:
        gp.layers.set_use_silu_activation(False)
    assert list(blk.state_dict().keys()) == keys  # same names in the same order as the reference module
    params = R.variant_params(blk.state_dict(), seed, keys)
    p = {"processor_list.0." + k: t.clone().requires_grad_(True) for k, t in params.items()}
    x, e = R.randn((N, H), seed + 1).requires_grad_(True), R.randn((E, H), seed + 2).requires_grad_(True)
    pos = R.randn((N, 3), seed + 5, 0.3)
    phi = R.randn((N,), seed + 6) if v.get("phi") else None
    x2, e2 = O.graph_net_block(x, e, ei, p, "processor_list.0.", v["act"], variant=v["variant"], pos=pos, phi=phi)
    assert torch.equal(x2.detach(), g[name + ".blk.x_out"]) and torch.equal(e2.detach()[:32], g[name + ".blk.e_out.rows32"])
    ((x2 * R.randn((N, H), seed + 3)).sum() + (e2 * R.randn((E, H), seed + 4)).sum()).backward()
    assert torch.allclose(x.grad, g[name + ".blk.dx"], rtol=1e-5, atol=1e-6)
    for k in keys:
        gr = p["processor_list.0." + k].grad
        if f"{name}.blk.g.{k}" in g:
            assert torch.allclose(gr, g[f"{name}.blk.g.{k}"], rtol=1e-5, atol=1e-6), k
        elif f"{name}.blk.g.{k}.norm" in g:
            assert abs(float(gr.norm()) - float(g[f"{name}.blk.g.{k}.norm"])) < 1e-5 * float(gr.norm()), k


RMSNORM_CASES = {"partial": dict(d=64, p=0.25, eps=1e-8, bias=False), "biased": dict(d=48, p=-1.0, eps=1e-8, bias=True),
                 "partial_biased_eps": dict(d=128, p=0.5, eps=1e-5, bias=True)}


def rmsnorm_case(name):
    """inputs of tests/golden/make_golden_rmsnorm.py (same numpy streams) + the reference's outputs"""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rmsnorm_variants.npz"))
    kw = RMSNORM_CASES[name]
    i = list(RMSNORM_CASES).index(name)
    d = kw["d"]
    return kw, R.randn((d,), 900 + i) * 0.3 + 1.0, R.randn((d,), 910 + i) * 0.2, R.randn((37, d), 920 + i), R.randn((37, d), 930 + i), \
        {k.split(".", 1)[1]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(name + ".")}


@pytest.mark.parametrize("name", list(RMSNORM_CASES))
def test_rmsnorm_variants_oracle_vs_golden(name):
    """partial / biased RMSNorm (layers.py:104-129): the oracle's restatement against the reference-minted fixture"""
    kw, scale, offset, x, cot, want = rmsnorm_case(name)
    xo, so = x.clone().requires_grad_(True), scale.clone().requires_grad_(True)
    y = O.rms_norm_general(xo, so, kw["d"], kw["p"], kw["eps"], offset if kw["bias"] else None)
    (y * cot).sum().backward()
    assert torch.equal(y.detach(), want["y"])
    assert torch.allclose(xo.grad, want["dx"], rtol=1e-6, atol=1e-7) and torch.allclose(so.grad, want["dscale"], rtol=1e-6, atol=1e-6)
