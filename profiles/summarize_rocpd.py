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
