#!/bin/bash
# weight-gradient kernels of a round on the bench shape, alternating in ONE GPU call: k_wgrad_pc (default) against k_wgrad_x6<6>
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  MGN_WGRAD_PC=1 python tools/kbench_wgrad.py ${1:-16} | sed 's/^/pc  /'
  MGN_WGRAD_PC=0 python tools/kbench_wgrad.py ${1:-16} | sed 's/^/x6  /'
done
