#!/bin/bash
# batch-1 (latency regime) A/B of environments on the CURRENT build: tools/ab_batch1_env.sh "ENV_A" "ENV_B" ...   (REPS, default 2)
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    env $v python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-c4 --no-kernel-timing --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('== $v: %.1f steps/s  %.3f ms/step  rollout %.3f ms  %s' % (d['value'], d['ms_per_step'], d['rollout_ms_per_step'], d['config'].get('launch')))"
  done
done
