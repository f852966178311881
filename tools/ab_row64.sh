#!/bin/bash
# row-vector weight gradients: split-bf16 form (default) against the exact-fp32 form (MGN_WGRAD_ROW64_EXACT=1), one GPU call:
# the c5 Transformer step and the shipped cylinder.json step
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for e in "" "MGN_WGRAD_ROW64_EXACT=1"; do
    echo "== ${e:-split-bf16}"
    env $e python tools/c5_modes.py 2>&1 | grep "^fp32" | head -1
    env $e python tools/shipped_modes.py 32 5 2>&1 | grep "graph train"
  done
done
