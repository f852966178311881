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
# (512 persistent workgroups) and the node-row launches of the same kernel differ 5x in duration
if len(sys.argv) > 3 and sys.argv[3] == "by-grid":
    print()
    print("kernel,workgroups,calls,avg_us,min_us,max_us")
    q = ("select name, grid_x/workgroup_x, count(*), avg(duration)/1000., min(duration)/1000., max(duration)/1000. "
         "from kernels where name like '%k_mlp_%' or name like '%k_wgrad%' or name like '%k_segsum%' "
         "group by name, grid_x/workgroup_x order by 4*count(*) desc")
    for name, wgs, calls, avg, mn, mx in db.execute(q):
        short = name.split("(")[0].replace("void ", "")
        print(f"\"{short}\",{wgs},{calls},{avg:.3f},{mn:.3f},{mx:.3f}")
