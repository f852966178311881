#!/bin/bash
# weight gradients of several rounds per launch (default) against one launch per round (MGN_WGRAD_BATCH_MB=0), one GPU call
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for e in "MGN_WGRAD_BATCH_MB=1024" "MGN_WGRAD_BATCH_MB=0"; do
    echo "== $e"
    env $e python tools/batch1_trace.py 2>&1 | tail -1
    env $e python bench.py --no-cpu-baseline --no-c4 --no-extras --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['roofline_other_kernels']
print('   batch16: %.2f steps/s  %.3f ms/step  wgrad call %.1f us' % (d['value'], d['ms_per_step'], o[1]['launch_ms']*1e3))"
  done
done
