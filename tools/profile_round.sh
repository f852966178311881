#!/bin/bash
# rocprofv3 evidence of a round, on the GPU box (run from the repo root through gpurun): kernel trace of the bench command,
# FETCH_SIZE / WRITE_SIZE passes over real training steps (separate runs: counters only), stall-attribution passes.
# Summaries are made afterwards from the merged gpurun_out/ (profiles/summarize_rocpd.py, tools/pmc_traffic_summary.py).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_${1:-r05}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 "$R/bench.py" --no-cpu-baseline --no-c4 --no-extras --steps 40 --warmup 8 > "$OUT/kt_bench.json" 2> "$OUT/kt.err"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- python3 "$R/tools/pmc_step.py" > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- python3 "$R/tools/pmc_step.py" > "$OUT/write.log" 2>&1
if [ "${2:-stall}" = "stall" ]; then bash "$R/tools/pmc_stall_passes.sh" "$OUT/stall" > "$OUT/stall.log" 2>&1; fi
# summaries on the box (gpurun merges at most 64 MiB back): the raw databases / per-dispatch CSVs stay behind
TAG=${1:-r05}
DB=$(find "$OUT/kt" -name "*.db" | head -1)
[ -n "$DB" ] && python3 "$R/profiles/summarize_rocpd.py" "$DB" 40 by-grid > "$OUT/${TAG}_bench_kernel_stats.csv"
C4_TRAFFIC_CSV="$R/profiles/${TAG}_c4_pmc_hbm_traffic.csv" PROFILES_DIR="$OUT" python3 "$R/tools/pmc_traffic_summary.py" "$OUT/fetch" "$OUT/write" "$TAG" > "$OUT/traffic_summary.log" 2>&1
[ -d "$OUT/stall" ] && python3 "$R/tools/pmc_stall_summary.py" "$OUT/stall" "$OUT/${TAG}_pmc_stall_counters.csv" > "$OUT/stall_summary.log" 2>&1
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/stall"
du -sh "$OUT"; ls "$OUT"; tail -2 "$OUT/kt_bench.json" | cut -c1-300
