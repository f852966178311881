#!/bin/bash
# per-launch durations, in launch order, of one configs[4] training step's Transformer block (rocprofv3 kernel trace of
# tools/c5_trace.py): which of the call sites of a kernel is the slow one.  usage (GPU box): tools/c5_launch_list.sh [fp32|bf16]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/c5_launches; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/kt" -o c5 -- python3 "$R/tools/c5_trace.py" ${1:-fp32} > "$OUT/kt.log" 2>&1
DB=$(find "$OUT/kt" -name "*.db" | head -1)
python3 - "$DB" > "$OUT/launches.txt" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, duration/1000. from kernels order by start"))
# the last training step: from the last k_clip_adamw_t backwards to the one before
idx = [i for i, r in enumerate(rows) if "k_clip_adamw_t" in r[0]]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = rows[lo][1]
for name, start, d in rows[lo:hi]:
    print(f"{(start - t0) / 1000.:10.1f} us  {d:8.1f} us  {name.split('(')[0].replace('void ', '')[:80]}")
PY
rm -rf "$OUT/kt"
head -150 "$OUT/launches.txt"
