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
