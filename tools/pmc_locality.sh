#!/bin/bash
# FETCH_SIZE of the edge kernels at the bench configuration: generator numbering vs per-mesh Morton numbering (tools/ab_b16_locality.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in raw loc; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/loc_$v --output-format csv -- python3 "$R/tools/ab_b16_locality.py" $v > "$R/gpurun_out/loc_$v.log" 2>&1; echo "$v rc=$?"
done
python3 - <<'P' | tee "$R/gpurun_out/pmc_locality.txt"
import csv, glob
from collections import defaultdict
print("# rocprofv3 --pmc FETCH_SIZE over tools/ab_b16_locality.py <raw|loc>; KB as reported (x2 for bytes, profiles/r03_pmc_calibration.txt)")
print("numbering,kernel,workgroups,launches,avg_KB_reported,avg_MB_corrected")
for v in ("raw", "loc"):
    acc = defaultdict(list)
    for f in glob.glob(f"/tmp/loc_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if k.startswith(("k_mlp_fwd_x6", "k_mlp_bwd_x6", "k_wgrad_x6", "k_segsum2")):
                    acc[(k, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))].append(float(r["Counter_Value"]))
    for (k, g), vals in sorted(acc.items()):
        a = sum(vals) / len(vals)
        print(f"{v},\"{k}\",{g},{len(vals)},{a:.1f},{a * 2 * 1024 / 1e6:.1f}")
P
