#!/bin/bash
# timing-only experiments: swap the engine library for a variant and time the edge kernel
cd "$(dirname "$0")/.."
cp graph-physics_amd/csrc/libmgn_hip.so /tmp/libmgn_orig.so
for v in "$@"; do
  cp tools/libexp_$v.so graph-physics_amd/csrc/libmgn_hip.so
  touch graph-physics_amd/csrc/libmgn_hip.so
  echo "== $v"; python tools/kbench.py --what ${WHAT:-edge_nosave} 2>&1 | grep -v amdgpu | tail -2
done
cp /tmp/libmgn_orig.so graph-physics_amd/csrc/libmgn_hip.so
