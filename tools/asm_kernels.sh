#!/bin/bash
# device assembly of csrc/mgn_kernels.hip -> /tmp/mgn_kernels.s (+ resource usage remarks in /tmp/mgn_kernels.rep); $1 = a kernel's mangled-name
# substring: its body is extracted to /tmp/k.s
cd "$(dirname "$0")/../graph-physics_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -I../../include -S --cuda-device-only -o /tmp/mgn_kernels.s mgn_kernels.hip -Rpass-analysis=kernel-resource-usage "${@:2}" > /tmp/mgn_kernels.rep 2>&1
grep -i "error" /tmp/mgn_kernels.rep | head
if [ -n "$1" ]; then
  awk -v pat="$1" '$0 ~ "^_Z[A-Za-z0-9_]*" pat "[A-Za-z0-9_]*:" {on=1} on {print} on && /s_endpgm/ {exit}' /tmp/mgn_kernels.s > /tmp/k.s
  grep -A9 "Function Name: _Z[A-Za-z0-9_]*$1" /tmp/mgn_kernels.rep | grep -E "VGPRs:|AGPRs|Scratch|Occupancy" | sed 's/.*remark: *//;s/ *\[-Rpass.*//' | paste - - - -
  wc -l /tmp/k.s
fi
