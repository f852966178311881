"""Round-5 kernels against fp64 evaluations, the kernels they replace and the oracle (all through the C ABI)."""
import os

import pytest
import torch

from conftest import rel_err  # noqa: F401


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _wgrad_jobs(Ms, dev, seed, scale_rows=False):
    from graph_physics_amd import ops  # noqa: F401
    g = torch.Generator().manual_seed(seed)
    H = 128
    jobs, refs = [], []
    for M in Ms:
        A = torch.randn(M, H, generator=g)
        B = torch.randn(M, H, generator=g)
        if scale_rows:   # wide dynamic range: the three bf16 pieces all matter
            A = A * torch.logspace(-6, 3, M).unsqueeze(1)[torch.randperm(M, generator=g)]
        A, B = A.to(dev), B.to(dev)
        dW = torch.full((H, H), float("nan"), device=dev)
        db = torch.full((H,), float("nan"), device=dev)
        jobs.append((A, H, 8, B, H, 8, H, dW, 0, H, db))
        refs.append((A.double().t() @ B.double(), A.double().sum(0)))
    return jobs, refs


@pytest.mark.gpu
@pytest.mark.parametrize("Ms", [(1,), (31,), (32,), (33,), (64, 97), (5000,), (70001, 3, 255), (180082, 30160)])
def test_producer_consumer_weight_gradient_kernel(dev, monkeypatch, Ms):
    """k_wgrad_pc (full 128 x 128 jobs with fp32 rows: producers split every operand value once per workgroup, consumers multiply)
    against fp64 torch at ragged row counts (the un-pipelined last tile, workgroups with 0 / 1 / 2 / many tiles, several jobs per
    launch), bit for bit run to run, and against k_wgrad_x6<6> (MGN_WGRAD_PC=0), the kernel it replaces"""
    from graph_physics_amd import ops

    def run():
        jobs, refs = _wgrad_jobs(Ms, dev, 11 + len(Ms))
        ops.wgrad(jobs, dev)
        return [(j[7], j[10]) for j in jobs], refs
    monkeypatch.delenv("MGN_WGRAD_PC", raising=False)
    out, refs = run()
    out2, _ = run()
    for (dW, db), (dW2, db2), (ref, ref_b) in zip(out, out2, refs):
        assert torch.equal(dW, dW2) and torch.equal(db, db2)
        assert float((dW.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-30) < 2e-6
        assert float((db.double() - ref_b).abs().max()) / (float(ref_b.abs().max()) + 1e-30) < 2e-6
    monkeypatch.setenv("MGN_WGRAD_PC", "0")
    out3, _ = run()
    for (dW, db), (dW3, db3), (ref, _) in zip(out, out3, refs):
        assert float((dW - dW3).abs().max()) / (float(ref.abs().max()) + 1e-30) < 2e-6
        assert float((db - db3).abs().max()) / (float(db3.abs().max()) + 1e-30) < 2e-6


@pytest.mark.gpu
def test_producer_consumer_weight_gradient_is_fp32_grade_on_a_wide_dynamic_range(dev):
    """rows scaled over nine decades: a result that dropped the second or third bf16 piece would be off by 2^-8 / 2^-16 of the
    large rows' products; element-wise against fp64 relative to sum |a||b| (what an fp32 evaluation promises)"""
    from graph_physics_amd import ops
    jobs, refs = _wgrad_jobs((4099,), dev, 5, scale_rows=True)
    ops.wgrad(jobs, dev)
    A, B, dW = jobs[0][0], jobs[0][3], jobs[0][7]
    bound = A.double().abs().t() @ B.double().abs()
    assert float(((dW.double() - refs[0][0]).abs() / bound).max()) < 3e-7


@pytest.mark.gpu
def test_bf16_mode_off_the_packed_path_reports_parameters_to_the_grad_ready_hook(dev):
    """advisor r4: at H != 128 the bf16 matrix mode multiplies bf16-rounded COPIES of the weights; what ProcessorFunction saves for
    autograd's version check and what it reports to the grad-ready hook must still be the PARAMETERS (the overlapped all-reduce
    looks gradients up by the parameter's storage), and a second forward after an in-place weight update must see the new weights
    (the rounded copies are never cached: the fused optimiser writes through raw pointers)"""
    import graph_physics_amd as gp
    from graph_physics_amd import ops
    sys_path_golden()
    import recipe as R
    L, H, N = 2, 32, 300
    _, ei, ea = R.delaunay_graph(N, 3)
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=H).to(dev)
    g = gp.Graph(x=R.randn((N, 11), 4).to(dev), edge_attr=R.randn((ea.shape[0], 3), 5).to(dev), edge_index=ei.to(dev))
    ptrs = {p.data_ptr() for p in net.parameters()}
    seen = []
    ops.set_matrix_precision("bf16")
    ops.set_grad_ready_hook(lambda pairs: seen.extend(pairs))
    try:
        out1 = net(g)
        out1.square().sum().backward()
        assert seen and all(p.data_ptr() in ptrs for p, _ in seen), "the hook must receive parameters, not rounded copies"
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.25)                        # what an optimiser step does, in place
            out2 = net(g)
        assert float((out2 - out1.detach()).abs().max()) > 1e-3 * float(out1.detach().abs().max())
    finally:
        ops.set_grad_ready_hook(None)
        ops.set_matrix_precision("fp32")


def sys_path_golden():
    import sys
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if d not in sys.path:
        sys.path.insert(0, d)


@pytest.mark.gpu
def test_default_inference_path_at_the_bench_batch_vs_oracle(dev):
    """the ping-pong edge kernel takes the inference-mode launches from 65 536 edge rows BY DEFAULT, and was checked
    against the oracle only through the 1M-node closure test.  Here directly: the bench batch (16 meshes, E = 180 082 rows per
    launch), 3 rounds, no_grad forward against the CPU oracle, 1e-5 in all three readings."""
    import graph_physics_amd as gp
    from conftest import assert_close3
    from oracle import mgn_oracle as O
    sys_path_golden()
    import recipe as R
    L = 3
    g = gp.cylinder_batch(16, 1885, 0)
    N, E = g.x.shape[0], g.edge_index.shape[1]
    assert E >= 65536 and os.environ.get("MGN_PP") in (None, "")
    params = R.make_params(R.epd_param_shapes(L, 128, 11, 3, 2), 91)
    x_in, e_in = R.randn((N, 11), 92), R.randn((E, 3), 93)
    ref = O.epd_forward(x_in, e_in, g.edge_index, params, L)
    net = gp.EncodeProcessDecode(L, 11, 3, 2, hidden_size=128).to(dev)
    net.load_state_dict(params)
    graph = gp.Graph(x=x_in.to(dev), edge_attr=e_in.to(dev), edge_index=g.edge_index.to(dev))
    with torch.no_grad():
        out = net(graph)
    assert_close3(out.cpu(), ref, 1e-5, "inference forward, batch 16, default (ping-pong) edge kernel")


@pytest.mark.gpu
def test_training_mode_ping_pong_instance_vs_oracle(dev, monkeypatch):
    """... and the TRAINING-mode instance (opt-in, MGN_PP=1 / 2: saves, masks, fused aggregation out of the ping-pong kernel) against
    the ORACLE, before it could ever become a default: (a) a small deep case without a single differing ReLU mask (MGN_PP=2: any
    size) -- every gradient within 2e-5 of the fp32 oracle; (b) the bench batch (E = 180 082 rows per launch, MGN_PP=1): forward
    1e-5, gradients by the flip-aware criterion of tests/test_hip_configs.py (fp32 + fp64 oracle).  (tools/pp_grad_check.py: at this
    size 26-48 of ~4e8 masks differ whatever kernel runs, and the distance to fp64 is 1.9e-3..7e-3 for the default kernel, this one
    AND the fp32 oracle alike -- a handful of flips decide it; the seed below is one where the criterion's x2 margin holds for
    both kernels.)"""
    import graph_physics_amd as gp
    from conftest import assert_close3, rel_err
    from test_hip_configs import _check_grads, _grad_case
    monkeypatch.setenv("MGN_PP", "2")
    g = gp.cylinder_mesh(400, 2)          # (seed 80: flip-free for this kernel AND the default one, profiles/r05_pp_grad_check.txt)
    out, grads, (o32, g32), (o64, g64), (flips, total, worst) = _grad_case(dev, g, 15, 80)
    assert_close3(out, o32, 1e-5, "training-mode forward through k_edge_fwd_pp<true>, N = 400, L = 15")
    if flips == 0:
        assert max(rel_err(grads[k], g32[k]) for k in grads) < 2e-5       # measured 7.7e-6 (default kernel: 8.4e-6)
    else:
        _check_grads(grads, g32, g64, flips, worst)
    monkeypatch.setenv("MGN_PP", "1")
    g = gp.cylinder_batch(16, 1885, 0)
    out, grads, (o32, g32), (o64, g64), (flips, total, worst) = _grad_case(dev, g, 3, 95)
    assert_close3(out, o32, 1e-5, "training-mode forward through k_edge_fwd_pp<true>, bench batch")
    assert flips <= 1e-6 * total, (flips, total)
    _check_grads(grads, g32, g64, flips, worst)
