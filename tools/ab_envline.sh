#!/bin/bash
# A/B of environments on the CURRENT build in one gpurun call: tools/ab_envline.sh "ENV_A" "ENV_B" ...  (REPS, default 2)
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    env $v python bench.py --no-cpu-baseline --no-c4 --steps ${STEPS:-40} --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
o=d['roofline_other_kernels']
print('== $v: %.2f steps/s  %.3f ms/step  rollout %.3f ms  edge_fwd %.1f us | ' % (d['value'], d['ms_per_step'], d['rollout_ms_per_step'], d['roofline']['launch_ms']*1e3) + ' | '.join('%s %.1f us' % (x['kernel'][:18], x['launch_ms']*1e3) for x in o))"
  done
done
