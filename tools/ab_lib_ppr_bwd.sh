#!/bin/bash
# timing of tools/kbench_ppr_bwd.py (bench edge shape) for several builds of the engine in ONE GPU call: tools/ab_lib_ppr_bwd.sh NAME...
cd "$(dirname "$0")/.."
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = base ]; then unset MGN_LIB; else export MGN_LIB=$PWD/tools/libexp_$n.so; fi
  echo "== $n (rep $rep)"
  timeout 300 python tools/kbench_ppr_bwd.py 16 none 2>&1 | grep -E "^ppr|^x6"
done
done
