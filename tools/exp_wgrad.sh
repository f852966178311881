#!/bin/bash
# timing experiments on k_wgrad_pc (variant builds tools/libexp_<TAG>.so, results invalid by design), two rounds in one GPU call
cd "$(dirname "$0")/.."
for rep in 1 2; do
  python tools/kbench_wgrad.py 16
  for v in "$@"; do MGN_LIB=tools/libexp_$v.so python tools/kbench_wgrad.py 16; done
done 2>&1 | grep "us"
