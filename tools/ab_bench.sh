#!/bin/bash
# A/B of two builds of the engine library in ONE gpurun call (box-to-box variation is ~2 %):
#   tools/ab_bench.sh OLD NEW   -> rollout ms/step and training steps/s for tools/libexp_<tag>.so, interleaved twice
cd "$(dirname "$0")/.."
cp graph-physics_amd/csrc/libmgn_hip.so /tmp/libmgn_orig.so
for rep in 1 2; do
  for v in "$@"; do
    cp tools/libexp_$v.so graph-physics_amd/csrc/libmgn_hip.so; touch graph-physics_amd/csrc/libmgn_hip.so
    echo "== $v: $(python bench.py --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | grep -o '"value[^,]*\|rollout_ms[^,]*' | tr '\n' ' ')"
  done
done
cp /tmp/libmgn_orig.so graph-physics_amd/csrc/libmgn_hip.so
