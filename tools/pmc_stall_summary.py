#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_stall_passes.sh (launches of the training steps,
keyed by kernel name + workgroup count).  python tools/pmc_stall_summary.py gpurun_out/pmc_stall [out.csv]"""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if name.startswith("void "): name = name[5:]
        if not name.startswith(("k_mlp", "k_wgrad", "k_edge", "k_segsum")): continue
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        acc[(name, wgs)][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (name, wgs), cs in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    n = len(next(iter(cs.values())))
    if n < 6: continue
    rows.append((name, wgs, n, {c: sum(v) / len(v) for c, v in cs.items()}))
allc = sorted({c for r in rows for c in r[3]})
out = sys.argv[2] if len(sys.argv) > 2 else None
lines = ["kernel,workgroups,launches," + ",".join(allc)]
for name, wgs, n, m in rows:
    lines.append('"%s",%d,%d,' % (name, wgs, n) + ",".join("%.4g" % m.get(c, float("nan")) for c in allc))
text = "\n".join(lines)
if out: open(out, "w").write(text + "\n")
for name, wgs, n, m in rows[:6]:
    print("== %s  (%d workgroups, %d launches)" % (name, wgs, n))
    for c in allc:
        if c in m: print("   %-40s %14.4g" % (c, m[c]))
