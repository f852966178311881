#!/bin/bash
# FETCH_SIZE / WRITE_SIZE factors of the engine's access patterns (tools/pmc_calib_probe.hip) -> gpurun_out/pmc_calib.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 150 rocprofv3 --pmc $c -d /tmp/calib_$c --output-format csv -- "$R/tools/pmc_calib_probe" > "$R/gpurun_out/calib_$c.log" 2>&1; echo "$c rc=$?"
done
python3 - <<'P' | tee "$R/gpurun_out/pmc_calib.txt"
import csv, glob
from collections import defaultdict
print("# tools/pmc_calib_probe under rocprofv3 --pmc (separate passes): every kernel touches 1 GiB exactly once; counter unit KB")
print("kernel,counter,launches,avg_KB,factor_vs_1GiB")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(list)
    for f in glob.glob(f"/tmp/calib_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        a = sum(v) / len(v)
        print(f"{k},{c},{len(v)},{a:.1f},{a * 1024 / 2**30:.4f}")
P
