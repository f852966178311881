#!/bin/bash
# rocprofv3 kernel trace of configs[4] training steps (tools/c5_trace.py) in both matrix modes, summarised per kernel.
# Run on the GPU box from the repo root through gpurun; copies of the two CSVs go to profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
OUT=$R/gpurun_out/prof_c5_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in fp32 bf16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$mode" -o c5 -- python3 "$R/tools/c5_trace.py" $mode > "$OUT/kt_$mode.log" 2>&1
  DB=$(find "$OUT/kt_$mode" -name "*.db" | head -1)
  [ -n "$DB" ] && python3 "$R/profiles/summarize_rocpd.py" "$DB" 40 > "$OUT/${TAG}_c5_${mode}_kernel_stats.csv"
  rm -rf "$OUT/kt_$mode"
done
ls "$OUT"; head -12 "$OUT/${TAG}_c5_fp32_kernel_stats.csv"
