"""ISA check of the static-shape chain kernels (csrc/mgn_x6.inc): their operand loads are inline asm that hipcc
does not track, ordered by the counted drains of the weight DMA.  The device assembly must not touch a load's
destination registers before the second counted drain after it (tools/check_untracked_loads.py)."""
import os, shutil, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


@pytest.mark.skipif(HIPCC is None, reason="hipcc not available")
def test_no_instruction_touches_an_untracked_load_before_its_drain(tmp_path):
    sys.path.insert(0, ROOT)
    from graph_physics_amd import _capi   # the code-generation flags of the shipped build (no packed-fp32 instructions)

    asm = tmp_path / "mgn.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17"] + list(_capi.DEVICE_FLAGS) +
                       ["-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", str(asm),
                        os.path.join(ROOT, "graph-physics_amd", "csrc", "mgn_kernels.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_untracked_loads.py"), str(asm)], capture_output=True, text=True)
    assert c.returncode == 0, c.stdout[-3000:]
    last = c.stdout.strip().splitlines()[-1]
    n = int(last.split()[0])
    assert n >= 100 and last.endswith("0 violations"), last  # the static kernels are there and clean
