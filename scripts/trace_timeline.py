#!/usr/bin/env python3
"""The last burst of kernels in a rocprofv3 kernel trace (bursts are separated by >= 20 ms of idle GPU), kernel by kernel:
start offset, duration, idle gap before it.  usage: trace_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        name = name[name.find("k_"):].split("(")[0] if "k_" in name else name.split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:60]))
rows.sort()
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 20_000_000:
        cut = i
rows = rows[cut:]
t0 = rows[0][0]
end = t0
busy = 0
for s, e, n in rows:
    print("%9.1f us  %8.1f us  gap %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3, n))
    busy += e - s
    end = max(end, e)
print("burst: %.3f ms wall, %.3f ms of kernels, %d kernels" % ((end - t0) / 1e6, busy / 1e6, len(rows)))
