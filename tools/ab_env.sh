#!/bin/bash
# A/B of one engine build under two environments in ONE gpurun call:  tools/ab_env.sh LIBTAG "ENV_A" "ENV_B" ...
cd "$(dirname "$0")/.."
lib=$1; shift
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    env $v MGN_LIB=tools/libexp_$lib.so python bench.py --no-cpu-baseline --no-c4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
o=d['roofline_other_kernels']
print('== $v: %.2f steps/s  rollout %.3f ms  edge_fwd %.1f us  bwd %.1f us  wgrad %.1f us  infer %.1f us' % (d['value'], d['rollout_ms_per_step'], d['roofline']['launch_ms']*1e3, o[0]['launch_ms']*1e3, o[1]['launch_ms']*1e3, o[2]['launch_ms']*1e3))"
  done
done
