#!/bin/bash
# rocprofv3 kernel trace of configs[2]-style training steps (tools/plate_trace.py: 16 plate meshes, bf16 matrix mode), per kernel.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r05}
OUT=$R/gpurun_out/prof_plate_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o plate -- python3 "$R/tools/plate_trace.py" > "$OUT/kt.log" 2>&1
DB=$(find "$OUT/kt" -name "*.db" | head -1)
[ -n "$DB" ] && python3 "$R/profiles/summarize_rocpd.py" "$DB" 40 by-grid > "$OUT/${TAG}_plate_kernel_stats.csv"
rm -rf "$OUT/kt"
tail -1 "$OUT/kt.log"; head -24 "$OUT/${TAG}_plate_kernel_stats.csv"
