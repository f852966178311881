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


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof of every argument struct as gcc lays out include/mgn_hip.h (LP64) against
    the ctypes mirrors in _capi.py."""
    import shutil
    import subprocess

    from graph_physics_amd import _capi as c

    structs = {"mgn_mlp_fwd_args": c.MlpFwdArgs, "mgn_mlp_bwd_args": c.MlpBwdArgs, "mgn_wgrad_job": c.WgradJob,
               "mgn_tblock": c.TBlock, "mgn_wpack_block": c.WpackBlock, "mgn_edge_bwd_fused_args": c.EdgeBwdFusedArgs, "mgn_linear_args": c.LinearArgs, "mgn_rownorm_phase": c.RownormPhase}
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "a C compiler is part of the toolchain contract"
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "mgn_hip.h"', "int main(void) {"]
    for name, st in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for fld, _ in st._fields_:
            lines.append(f'printf("{name}.{fld} %zu\\n", offsetof({name}, {fld}));')
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([cc, "-I", os.path.join(REPO, "include"), "-o", str(exe), str(src)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, st in structs.items():
        assert int(got[name]) == ctypes.sizeof(st), name
        for fld, _ in st._fields_:
            assert int(got[f"{name}.{fld}"]) == getattr(st, fld).offset, f"{name}.{fld}"
    assert c.MAX_LAYERS == 8 and c.MAX_PHASES == 3 and c.MAX_WGRAD_JOBS == 48
    assert c.WPACK_BYTES == 98304


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


def test_two_byte_operands_are_refused_outside_the_bf16_packed_path():
    """[r4] negative leading dimensions of mgn_wgrad_job (two-byte rows) and precision 2 of mgn_mlp_fwd are taken only where a kernel
    reads / writes that form: validated before any launch, so this runs without a GPU."""
    from graph_physics_amd import _capi

    lib = _capi.lib()
    job = (_capi.WgradJob * 1)()
    job[0].M, job[0].lda, job[0].ldb, job[0].ldw, job[0].nja, job[0].nkb, job[0].kw = 1000, -128, 128, 128, 8, 8, 128
    assert lib.mgn_wgrad_p(1, job, None, 0, 0, None) == 1           # fp32-grade precision: no two-byte operand
    assert b"negative leading dimension" in lib.mgn_last_error()
    job[0].lda, job[0].nja = -64, 4                                   # not a full 128 x 128 job
    assert lib.mgn_wgrad_p(1, job, None, 0, 1, None) == 1
    a = _capi.MlpFwdArgs()
    a.M, a.H, a.NL, a.nphase, a.out_w, a.precision = 100, 128, 4, 1, 128, 2   # no packed weights: not the split-bf16 path
    a.kw[0] = 128
    assert lib.mgn_mlp_fwd(ctypes.byref(a), None) == 1
    assert b"precision 2" in lib.mgn_last_error()
    a.precision = 3
    assert lib.mgn_mlp_fwd(ctypes.byref(a), None) == 1
