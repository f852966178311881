#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 --pmc counter_collection CSVs.
usage: pmc_summary.py <dir-or-csv> [...]  (kernels whose name contains 'x6', 'lds', 'segsum')"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for a in sys.argv[1:]:
    files = [a] if a.endswith(".csv") else glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not any(t in k for t in ("x6", "_lds", "segsum")):
                continue
            key = (k[:40], r["Grid_Size"])
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(acc):
    print(key)
    for c in sorted(acc[key]):
        v = acc[key][c]
        print(f"    {c:32s} n={len(v):3d} avg {sum(v)/len(v):16.1f}")
