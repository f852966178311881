#!/usr/bin/env python3
"""Dump the per-kernel summary (calls, total, average, share) of a rocprofv3 rocpd
SQLite database -- what `rocprofv3 --kernel-trace --stats` collected -- as CSV.
usage: summarize_rocpd.py <results.db> [max_rows] > profiles/<name>.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
print("kernel,calls,total_us,avg_us,percent")
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit ?", (n,)):
    short = name.split("(")[0].replace("void ", "")
    if len(short) > 90:
        short = short[:87] + "..."
    print(f"\"{short}\",{calls},{tot:.1f},{avg:.3f},{pct:.2f}")

# per launch shape (grid size in workgroups) for the engine kernels: the edge-row launches
# (512 persistent workgroups) and the node-row launches of the same kernel differ 5x in duration;
# the median separates the training-mode launches (the majority) from the few inference-mode
# launches of the rollout part of the same command
if len(sys.argv) > 3 and sys.argv[3] == "by-grid":
    import statistics
    print()
    print("kernel,workgroups,calls,avg_us,median_us,min_us,max_us")
    groups = {}
    q = ("select name, grid_x/workgroup_x, duration/1000. from kernels where name like '%k_mlp_%' or name like '%k_wgrad%' "
         "or name like '%k_segsum%' or name like '%k_seg_fix%' or name like '%k_clip_adamw%' or name like '%k_sumsq%'")
    for name, wgs, d in db.execute(q):
        groups.setdefault((name.split("(")[0].replace("void ", ""), wgs), []).append(d)
    for (short, wgs), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        print(f"\"{short}\",{wgs},{len(v)},{sum(v)/len(v):.3f},{statistics.median(v):.3f},{min(v):.3f},{max(v):.3f}")
