#!/bin/bash
# batch-1 (latency regime) A/B under environment variants:  tools/ab_batch1.sh LIBTAG "ENV_A" "ENV_B" ...
cd "$(dirname "$0")/.."
lib=$1; shift
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    env $v MGN_LIB=tools/libexp_$lib.so python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-c4 --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('== $v: %.1f steps/s  %.3f ms/step  rollout %.3f ms' % (d['value'], d['ms_per_step'], d['rollout_ms_per_step']))"
  done
done
