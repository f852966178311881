"""The sparse-attention Transformer path (BASELINE.json configs[4]) and the temporal block.
CPU: the oracle against the fixtures minted from the reference over a dgl.sparse stand-in
(tests/golden/make_golden_transformer.py).  GPU: the HIP sparse-attention kernels against the oracle
(ragged rows, duplicate edges, self loops, empty rows, 1..16 heads, all widths; forward + q/k/v gradients),
and the product modules through the JSON surface against the reference-minted fixtures."""
import os

import numpy as np
import pytest
import torch

import recipe as R
from conftest import assert_close3, rel_err
from oracle import mgn_oracle as O

FWD_TOL, GRAD_TOL = 1e-5, 1e-4


def fixture():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transformer.npz"))
    return {k: (str(z[k]) if z[k].dtype.kind == "U" else torch.from_numpy(z[k])) for k in z.files}


def _case(name):
    import graph_physics_amd as gp

    c = R.TRANSFORMER_CASES[name]
    H, nh, L, N, seed = c["hidden"], c["heads"], c["L"], c["N"], c["seed"]
    fin, fout = c.get("f_in", 11), c.get("out", 2)
    x_in, cot = R.randn((N, fin), seed + 1), R.randn((N, fout), seed + 3)
    if c["model"] == "etd":
        pos, ei, _ = R.delaunay_graph(N, seed, dim=c.get("pos_dim", 3))
        cfg = {"model": {"type": "transformer", "message_passing_num": L, "hidden_size": H, "node_input_size": fin - 9, "output_size": fout,
                         "edge_input_size": 0, "num_heads": nh, "use_rope_embeddings": c.get("rope", False),
                         "use_gated_attention": c.get("gate", False), "rope_pos_dimension": c.get("pos_dim", 3)},
               "training": {"use_temporal_block": c.get("temporal", False)}}
        net = gp.get_model(cfg)
        e_in = None
    else:
        pos, ei, ea = R.delaunay_graph(N, seed, dim=2)
        net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=H, use_temporal_block=True)
        e_in = R.randn((ea.shape[0], 3), seed + 2)
    return c, net, pos, ei, x_in, e_in, cot


@pytest.mark.parametrize("name", list(R.TRANSFORMER_CASES))
def test_oracle_vs_reference_golden(name):
    fx = fixture()
    c, net, pos, ei, x_in, e_in, cot = _case(name)
    keys = fx[name + ".keys"].split("|")
    assert list(net.state_dict().keys()) == keys     # the product module has the reference's parameter names / order
    params = R.variant_params(net.state_dict(), c["seed"], keys)
    p = {k: t.clone().requires_grad_(t.is_floating_point() and not k.endswith("inv_freq")) for k, t in params.items()}
    if c["model"] == "etd":
        out = O.etd_forward(x_in, ei, p, c["L"], c["heads"], pos=pos, use_rope=c.get("rope", False), use_gate=c.get("gate", False),
                            pos_dimension=c.get("pos_dim", 3), use_temporal_block=c.get("temporal", False))
    else:
        x, e = O.mlp(x_in, p, "nodes_encoder."), O.mlp(e_in, p, "edges_encoder.")
        prev = x
        for i in range(c["L"]):
            prev = x
            x, e = O.graph_net_block(x, e, ei, p, f"processor_list.{i}.")
        out = O.mlp(O.temporal_attention(prev, x, p, "temporal_block.", ei, 4), p, "decode_module.")
    assert rel_err(out, fx[name + ".out"]) < 2e-6
    (out * cot).sum().backward()
    for k in keys:
        if f"{name}.gnorm.{k}" in fx and p[k].grad is not None:
            gn = float(fx[f"{name}.gnorm.{k}"])
            if gn < 1e-4:   # without RoPE a key bias shifts every score of a row alike: softmax-invariant, gradient = rounding noise
                assert k.endswith("k_proj.bias") and float(p[k].grad.norm()) < 1e-4, k
                continue
            # ReLU encoders / decoder / blocks: a pre-activation at rounding distance from zero can take the other
            # branch on another host CPU (thread count changes the GEMM summation order): norms within 2e-3
            assert abs(float(p[k].grad.norm()) - gn) < 2e-3 * gn + 1e-8, k


# ----------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("H,nh", [(128, 4), (64, 4), (64, 2), (32, 8), (16, 16), (128, 1), (128, 16), (16, 1)])
def test_sparse_attention_kernels_vs_oracle(dev, H, nh):
    from graph_physics_amd import transformer as T

    N, E, seed = 300, 2500, 7 + H + nh
    ei = R.random_graph(N, E, seed)            # unsorted, duplicates, self loops, node N-1 never a source nor a target
    ei[0, 100:140] = 17                        # a hub row
    q, k, v, cot = (R.randn((N, H), seed + i) for i in range(1, 5))
    D = H // nh
    qo, ko, vo = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = O.sparse_attention(qo.reshape(N, D, nh), ko.reshape(N, D, nh), vo.reshape(N, D, nh), ei).reshape(N, H)
    (ref * cot).sum().backward()
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    topo = T.AttnTopology(ei.to(dev), N)
    y = T.sparse_attention(qd, kd, vd, topo, nh)
    (y * cot.to(dev)).sum().backward()
    assert_close3(y, ref, FWD_TOL, f"attention H={H} heads={nh}")
    rows = torch.unique(ei[0])
    empty = np.setdiff1d(np.arange(N), rows.numpy())
    assert empty.size > 0 and float(y[torch.from_numpy(empty).to(dev)].abs().max()) == 0.0   # rows without edges: zeros
    for a, b, what in ((qd.grad, qo.grad, "dq"), (kd.grad, ko.grad, "dk"), (vd.grad, vo.grad, "dv")):
        assert rel_err(a, b) < 2e-5, what
    # deterministic (CSR order, no atomics)
    qd.grad = kd.grad = vd.grad = None
    y2 = T.sparse_attention(qd, kd, vd, topo, nh)
    (y2 * cot.to(dev)).sum().backward()
    assert torch.equal(y, y2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(R.TRANSFORMER_CASES))
def test_transformer_models_vs_reference_golden(dev, name):
    import graph_physics_amd as gp

    fx = fixture()
    c, net, pos, ei, x_in, e_in, cot = _case(name)
    net = net.to(dev)
    net.load_state_dict(R.variant_params(net.state_dict(), c["seed"], fx[name + ".keys"].split("|")))
    g = gp.Graph(x=x_in.to(dev), edge_index=ei.to(dev), pos=pos.to(dev))
    if e_in is not None:
        g.edge_attr = e_in.to(dev)
    out = net(g)
    assert_close3(out, fx[name + ".out"], FWD_TOL, name)
    (out * cot.to(dev)).sum().backward()
    for k, p in net.state_dict(keep_vars=True).items():
        if f"{name}.gnorm.{k}" in fx:
            gn = float(fx[f"{name}.gnorm.{k}"])
            if gn < 1e-4:                   # key bias without RoPE: mathematically zero (softmax invariance), noise on both sides
                assert float(p.grad.norm()) < 1e-4, k
                continue
            assert abs(float(p.grad.norm()) - gn) < 2e-3 * gn + 1e-7, k   # ReLU encoder / decoder: a flipped mask, see test_hip_variants
        if f"{name}.g.{k}" in fx:
            from conftest import rms_err
            assert rms_err(p.grad, fx[f"{name}.g.{k}"]) < 2e-3, k
