import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) where no device is visible, e.g. a plain `pytest tests/`
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b|  (relative to the tensor's scale)"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# Parity vocabulary.  north_star: "within 1e-5 relative fp32".  Three readings are checked:
#   rel_err   max|a-b| / max|b|            tensor-scale (the headline bar)
#   rms_err   ||a-b||_2 / ||b||_2          RMS-relative: no single element hides a drift
#   elem_err  max over {|b| > floor*max|b|} of |a-b| / |b|   element-wise, on the elements that carry
#             signal (default floor 1e-3 of the tensor's scale; below it only rel_err constrains)
def rms_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def elem_err(a: torch.Tensor, b: torch.Tensor, floor: float = 1e-3) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    keep = b.abs() > floor * b.abs().max()
    if not bool(keep.any()):
        return 0.0
    return float(((a - b).abs()[keep] / b.abs()[keep]).max())


def assert_close3(a, b, tol, what="", elem_factor=100.0):
    """the three readings at once: tensor-scale and RMS at `tol`; element-wise (|b| above 1e-3 of the
    scale) at elem_factor*tol -- an element 1000x below the scale carries 1000x the relative rounding"""
    r, q, e = rel_err(a, b), rms_err(a, b), elem_err(a, b)
    assert r < tol, f"{what}: max-relative {r:.3e} >= {tol}"
    assert q < tol, f"{what}: rms-relative {q:.3e} >= {tol}"
    assert e < elem_factor * tol, f"{what}: element-wise {e:.3e} >= {elem_factor * tol}"
    return r, q, e
