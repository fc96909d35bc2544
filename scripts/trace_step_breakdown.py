#!/usr/bin/env python3
"""Per-step breakdown of the sharded loop (pass -> all-gather -> solve) from a rocprofv3 kernel trace: over the last `steps`
occurrences of k_icp_solve, the mean time from one solve's end to the next (a step), the mean duration of each kernel inside a
step, and what is left (gaps).  usage: trace_step_breakdown.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        name = name[name.find("k_"):].split("(")[0] if "k_" in name else name.split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:70]))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
solves = [i for i, r in enumerate(rows) if r[2].startswith("k_icp_solve")]
# the longest run of consecutive steady steps at the end of the trace: skip the last registration's tail (finish kernels)
use = solves[-(steps + 1):]
tot = 0.0
per = defaultdict(float)
cnt = defaultdict(int)
for a, b in zip(use[:-1], use[1:]):
    t0, t1 = rows[a][1], rows[b][1]
    tot += t1 - t0
    for s, e, n in rows[a + 1:b + 1]:
        per[n] += e - s
        cnt[n] += 1
n = len(use) - 1
print("steps: %d, mean step %.2f us" % (n, tot / n / 1e3))
busy = 0.0
for k in sorted(per, key=lambda k: -per[k]):
    print("  %-70s %8.2f us per step (%.1f launches per step)" % (k, per[k] / n / 1e3, cnt[k] / n))
    busy += per[k]
print("  %-70s %8.2f us per step" % ("gaps between kernels", (tot - busy) / n / 1e3))
