#!/bin/bash
# A/B of engine-library builds in ONE gpurun call (box-to-box variation is ~2 %), through MGN_LIB:
#   tools/ab_lib.sh TAG1 TAG2 ...   -> headline steps/s, rollout ms and the chain kernels' in-situ launch times
#   for tools/libexp_<TAG>.so, interleaved REPS (default 2) times
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    MGN_LIB=tools/libexp_$v.so python bench.py --no-cpu-baseline --no-c4 ${AB_EXTRA:---no-extras} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
o=d['roofline_other_kernels']
print('== $v: %.2f steps/s  rollout %.3f ms  edge_fwd %.1f us  bwd %.1f us  wgrad %.1f us  infer %.1f us' % (d['value'], d['rollout_ms_per_step'], d['roofline']['launch_ms']*1e3, o[0]['launch_ms']*1e3, o[1]['launch_ms']*1e3, o[2]['launch_ms']*1e3))"
  done
done
