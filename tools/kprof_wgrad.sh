#!/bin/bash
# kernel-only durations of the weight-gradient launches of tools/kbench_wgrad.py under rocprofv3 (kernel trace), per kernel name
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/kprof_wgrad
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pc in 1 0; do
  MGN_WGRAD_PC=$pc timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt$pc" --output-format csv -- python3 "$R/tools/kbench_wgrad.py" 16 > "$OUT/log$pc" 2>&1
  f=$(find "$OUT/kt$pc" -name "*kernel_stats.csv" | head -1)
  echo "== MGN_WGRAD_PC=$pc"; grep -i "wgrad" "$f" | cut -c1-200
done
rm -rf "$OUT"/kt*
