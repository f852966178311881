"""GPU: the GraphNetBlock variants on the HIP engine against the fixtures minted from
the reference (tests/golden/make_golden_variants.py) -- SiLU, sigmoid gate (with / without phi), relative
RoPE (2 / 3 axes), gated-MLP blocks (GELU / SiLU), nb_of_layers=3 without RMSNorm, and the combination --
one block on the ragged multigraph with every gradient, and a 2-round EncodeProcessDecode through the
JSON surface (parse_parameters.get_model).  Forward 1e-5, gradients 1e-4 (the suite's bars)."""
import pytest
import torch

import recipe as R
import graph_physics_amd as gp
from conftest import assert_close3, rel_err
from test_oracle_golden import variant_fixture

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 1e-5, 1e-4
H = 128


@pytest.fixture(scope="module")
def fx():
    return variant_fixture()


@pytest.mark.parametrize("name", list(R.VARIANTS))
def test_block_variant_vs_reference_golden(dev, fx, name):
    v = R.VARIANTS[name]
    N, E, seed = 40, 150, 300 + v["seed"]
    ei = R.random_graph(N, E, seed)
    gp.layers.set_use_silu_activation(v["act"] == "silu")
    try:
        blk = gp.GraphNetBlock(H, **R.block_kwargs(v)).to(dev)
    finally:
        gp.layers.set_use_silu_activation(False)
    keys = fx[name + ".blk.keys"].split("|")
    blk.load_state_dict(R.variant_params(blk.state_dict(), seed, keys))
    x = R.randn((N, H), seed + 1).to(dev).requires_grad_(True)
    e = R.randn((E, H), seed + 2).to(dev).requires_grad_(True)
    pos = R.randn((N, 3), seed + 5, 0.3).to(dev)
    phi = R.randn((N,), seed + 6).to(dev) if v.get("phi") else None
    x2, e2 = blk(x, ei.to(dev), e, pos=pos if v["variant"].get("use_rope") else None, phi=phi)
    assert_close3(x2, fx[name + ".blk.x_out"], FWD_TOL, name + " x'")
    assert rel_err(e2[:32], fx[name + ".blk.e_out.rows32"]) < FWD_TOL
    assert abs(float(e2.norm()) - float(fx[name + ".blk.e_out.norm"])) < FWD_TOL * float(e2.norm())
    ((x2 * R.randn((N, H), seed + 3).to(dev)).sum() + (e2 * R.randn((E, H), seed + 4).to(dev)).sum()).backward()
    assert rel_err(x.grad, fx[name + ".blk.dx"]) < GRAD_TOL
    assert rel_err(e.grad[:32], fx[name + ".blk.de.rows32"]) < GRAD_TOL
    for k, p in blk.state_dict(keep_vars=True).items():
        if f"{name}.blk.g.{k}" in fx:
            assert rel_err(p.grad, fx[f"{name}.blk.g.{k}"]) < GRAD_TOL, k
        elif f"{name}.blk.g.{k}.norm" in fx:
            assert rel_err(p.grad[:4], fx[f"{name}.blk.g.{k}.rows4"]) < GRAD_TOL, k
            gn = float(fx[f"{name}.blk.g.{k}.norm"])
            assert abs(float(p.grad.norm()) - gn) < GRAD_TOL * gn, k
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k  # gate_pos without phi


@pytest.mark.parametrize("name", list(R.VARIANTS))
def test_epd_variant_through_json_vs_reference_golden(dev, fx, name):
    v = R.VARIANTS[name]
    vv = v["variant"]
    L, N2, seed2 = 2, 200, 400 + v["seed"]
    pos2, ei2, ea2 = R.delaunay_graph(N2, seed2, dim=vv.get("rope_axes", 3) if vv.get("use_rope") else 2)
    cfg = {"model": {"type": "epd", "message_passing_num": L, "hidden_size": H, "node_input_size": 2, "output_size": 2,
                     "edge_input_size": ea2.shape[1], "use_silu_activation": v["act"] == "silu",
                     "use_gated_mlp": vv.get("use_gated_mlp", False), "use_rope_embeddings": vv.get("use_rope", False),
                     "use_gated_attention": vv.get("use_gate", False), "rope_pos_dimension": vv.get("rope_axes", 3),
                     "rope_base": vv.get("rope_base", 10000.0)}, "training": {}}
    try:
        net = gp.get_model(cfg).to(dev)   # node_input_size 2 + 9 one-hot = 11
    finally:
        gp.layers.set_use_silu_activation(False)
    keys = fx[name + ".epd.keys"].split("|")
    net.load_state_dict(R.variant_params(net.state_dict(), seed2, keys))
    x_in, e_in = R.randn((N2, 11), seed2 + 1), R.randn((ea2.shape[0], ea2.shape[1]), seed2 + 2)
    g = gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=ei2.to(dev), pos=pos2.to(dev))
    if v.get("phi"):
        g.phi = R.randn((N2,), seed2 + 6).to(dev)
    out = net(g)
    assert_close3(out, fx[name + ".epd.out"], FWD_TOL, name + " EPD")
    (out * R.randn((N2, 2), seed2 + 3).to(dev)).sum().backward()
    for k, p in net.state_dict(keep_vars=True).items():
        if f"{name}.epd.gnorm.{k}" in fx:
            gn = float(fx[f"{name}.epd.gnorm.{k}"])
            assert abs(float(p.grad.norm()) - gn) < 2 * GRAD_TOL * gn + 1e-7, k
        if f"{name}.epd.g.{k}" in fx:
            assert rel_err(p.grad, fx[f"{name}.epd.g.{k}"]) < 2 * GRAD_TOL, k


def test_silu_narrow_and_generic_widths_vs_oracle(dev):
    """SiLU off the packed path too: hidden 32 (generic kernels) end to end against the oracle."""
    from oracle import mgn_oracle as O

    gp.layers.set_use_silu_activation(True)
    try:
        net = gp.EncodeProcessDecode(3, 11, 3, 2, hidden_size=32).to(dev)
    finally:
        gp.layers.set_use_silu_activation(False)
    N = 300
    _, ei, ea = R.delaunay_graph(N, 5)
    params = R.make_params(R.epd_param_shapes(3, 32, 11, 3, 2), 9)
    net.load_state_dict(params)
    x_in, e_in, cot = R.randn((N, 11), 1), R.randn((ei.shape[1], 3), 2), R.randn((N, 2), 3)
    p = {k: t.clone().requires_grad_(True) for k, t in params.items()}
    ref = O.epd_forward(x_in, e_in, ei, p, 3, act="silu")
    (ref * cot).sum().backward()
    out = net(gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=ei.to(dev)))
    (out * cot.to(dev)).sum().backward()
    assert_close3(out, ref, FWD_TOL, "silu H=32")
    for k, q in net.named_parameters():
        assert rel_err(q.grad, p[k].grad) < GRAD_TOL, k
