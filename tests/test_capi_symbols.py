"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mgn_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

from conftest import REPO


def _declared():
    src = open(os.path.join(REPO, "include", "mgn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgn_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    from graph_physics_amd import _capi

    lib = _capi.lib()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mgn_hip.h but not exported"
        assert n in _capi.SYMBOLS, f"{n} has no ctypes prototype"
    assert lib.mgn_version() >= 100
    assert lib.mgn_last_error() == b""


def test_struct_layout_matches_header():
    from graph_physics_amd import _capi as c

    # sizes the C compiler produces for the same field order (LP64)
    assert ctypes.sizeof(c.MlpFwdArgs) == 8 + 12 + 4 + 24 + 24 + 12 + 4 + 64 + 64 + 8 + 8 + 24 + 64 + 16 + 8 + 32 + 8 + 32
    assert ctypes.sizeof(c.WgradJob) == 24 + 8 + 24 + 8
    assert c.MAX_LAYERS == 8 and c.MAX_PHASES == 3 and c.MAX_WGRAD_JOBS == 12


def test_host_side_queries_need_no_gpu():
    from graph_physics_amd import _capi

    lib = _capi.lib()
    assert lib.mgn_csr_workspace_bytes(100, 10) >= 10 * 4
    assert lib.mgn_mlp_bwd_workspace_bytes(1000, 128, 4) > 0
    # argument validation happens before any launch
    a = _capi.MlpFwdArgs()
    a.M, a.H, a.NL, a.nphase, a.out_w = 0, 100, 4, 1, 100
    assert lib.mgn_mlp_fwd(ctypes.byref(a), None) == 1
    assert b"H must be" in lib.mgn_last_error()
