#!/usr/bin/env python3
"""Experiment: start the second dispatch round of the packed chain kernels (workgroups >= gridDim/2, which share
their CUs with the first round -- tools/timeline_x6.py census) `t` x 10 ns late, so the two waves of a SIMD are out
of phase.  Needs tools/libexp_STG.so (engine built with -DMGN_EXP_PAIR_STAGGER).  python tools/stagger_sweep.py 0 100 200"""
import io, json, os, sys, contextlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["MGN_LIB"] = os.path.join(R, "tools", "libexp_STG.so")
import importlib
capi = importlib.import_module("graph-physics_amd._capi")
L = capi.lib()
import bench
for t in [int(v) for v in sys.argv[1:]] or [0, 100, 200, 300]:
    assert L.mgn_debug_set_stagger(t) == 0
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-c4"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    o = d["roofline_other_kernels"]
    print("stagger %4d ticks: %.2f steps/s  rollout %.3f ms  edge_fwd %.1f us  bwd %.1f us  wgrad %.1f us  infer %.1f us" % (
        t, d["value"], d["rollout_ms_per_step"], d["roofline"]["launch_ms"] * 1e3, o[0]["launch_ms"] * 1e3,
        o[1]["launch_ms"] * 1e3, o[2]["launch_ms"] * 1e3), flush=True)
