#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv): how busy the GPU was and how many kernels ran at once.
usage: trace_concurrency.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = rows[0][0] + skip * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
ev = []
for s, e, _ in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = 0
level = 0
hist = defaultdict(int)
prev = ev[0][0]
for t, d in ev:
    if level > 0:
        busy += t - prev
    hist[level] += t - prev
    level += d
    prev = t
total = sum(e - s for s, e, _ in rows)
print("window %.1f ms  kernels %d  busy %.1f%%  sum of durations / window = %.2f" % (span / 1e6, len(rows), 100.0 * busy / span, total / span))
print("time share by number of kernels in flight:", {k: round(100.0 * v / span, 1) for k, v in sorted(hist.items())})
by = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    by[n][0] += 1
    by[n][1] += e - s
for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-62s calls %7d  mean %8.2f us  share of summed durations %5.1f%%" % (n, c, d / c / 1e3, 100.0 * d / total))
