#!/bin/bash
# timing of tools/kbench_ppr.py (bench edge shape, inference + training mode) for several builds of the engine in ONE GPU call:
#   tools/ab_lib_ppr.sh NAME...   (NAME = base for the shipped library, otherwise tools/libexp_NAME.so)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = base ]; then unset MGN_LIB; else export MGN_LIB=$PWD/tools/libexp_$n.so; fi
  echo "== $n (rep $rep)"
  timeout 300 python tools/kbench_ppr.py 16 none 2>&1 | grep -E "^ppr|^x6" | sort | uniq -c | awk '{print}' 
done
done
