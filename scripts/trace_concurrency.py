#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv): how busy the GPU was and how many kernels ran at once.
usage: trace_concurrency.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        name = name[name.find("k_"):].split("(")[0].split("<")[0] if "k_" in name else name.split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:60], r.get("Queue_Id", "?")))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = rows[0][0] + skip * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
ev = []
for s, e, *_ in rows:
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
total = sum(e - s for s, e, *_ in rows)
print("window %.1f ms  kernels %d  busy %.1f%%  sum of durations / window = %.2f" % (span / 1e6, len(rows), 100.0 * busy / span, total / span))
print("time share by number of kernels in flight:", {k: round(100.0 * v / span, 1) for k, v in sorted(hist.items())})
by = defaultdict(lambda: [0, 0])
for s, e, n, _ in rows:
    by[n][0] += 1
    by[n][1] += e - s
for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-62s calls %7d  mean %8.2f us  share of summed durations %5.1f%%" % (n, c, d / c / 1e3, 100.0 * d / total))

# per hardware queue: how much of the window it ran a kernel, and the gaps between consecutive kernels
perq = defaultdict(list)
for s, e, n, q in rows:
    perq[q].append((s, e))
for q, iv in sorted(perq.items()):
    iv.sort()
    busy_q = sum(e - s for s, e in iv)
    gaps = [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
    gaps_pos = sorted(g for g in gaps if g > 0)
    med = gaps_pos[len(gaps_pos) // 2] / 1e3 if gaps_pos else 0.0
    print("queue %s: %6d kernels, busy %5.1f%% of the window, gap between consecutive kernels: median %.1f us, mean %.1f us" %
          (q, len(iv), 100.0 * busy_q / span, med, (sum(gaps_pos) / max(len(gaps_pos), 1)) / 1e3))
