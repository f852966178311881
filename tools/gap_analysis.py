#!/usr/bin/env python3
"""Inter-kernel gaps of the training step from a rocprofv3 --kernel-trace CSV: for the last steps of tools/pmc_step.py, the GPU-busy time
(union of kernel intervals) against the wall time between the first and the last kernel, and the gaps by the kernel that follows them.
usage: python tools/gap_analysis.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]))
rows.sort()
# the training steps: from the third-last k_clip_adamw group to the end
# steps are delimited by the first k_sumsq_partial* launch of each optimiser tail (one launch, or several in a row)
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_sumsq_partial") and not (i > 0 and rows[i - 1][2].startswith("k_sumsq_partial"))]
a, b = starts[-3], starts[-1]          # two whole steps
seg = rows[a:b]
wall = seg[-1][1] - seg[0][0]
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
gaps = defaultdict(lambda: [0, 0])
for s, e, n in seg[1:]:
    if s > cur_e:
        gaps[n][0] += s - cur_e; gaps[n][1] += 1
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"two steps: wall {wall / 2e6:.3f} ms per step, GPU busy {busy / 2e6:.3f} ms per step, idle {(wall - busy) / 2e6:.3f} ms per step over {len(seg) // 2} launches per step")
for n, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  before {n:40s} {t / 2e3:8.1f} us per step in {c // 2:3d} gaps ({t / max(c, 1) / 1e3:5.1f} us each)")

per = defaultdict(lambda: [0, 0])
for st, en, n in seg:
    per[n][0] += en - st; per[n][1] += 1
print("kernel time per step (top 30):")
for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"  {n:40s} {t / 2e3:9.1f} us per step in {c / 2:6.1f} launches")
