#!/bin/bash
# build a variant of the engine library for timing experiments: tools/mkvar.sh NAME [extra hipcc flags]  ->  tools/libexp_NAME.so
cd "$(dirname "$0")/.."
name=$1; shift
SRCS="graph-physics_amd/csrc/mgn_kernels.hip graph-physics_amd/csrc/mgn_prep.hip graph-physics_amd/csrc/mgn_attn.hip graph-physics_amd/csrc/mgn_dense.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude "$@" -o tools/libexp_$name.so $SRCS 2>&1 | grep -E "error" | head
ls -la tools/libexp_$name.so
