#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv): the largest idle gaps of the GPU in the last part of the run, with the kernels either
side.  usage: trace_gaps.py <kernel_trace.csv> [skip_fraction] [how many]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        name = name[name.find("k_"):].split("(")[0] if "k_" in name else name.split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:48]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.75
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
t_lo = rows[0][0] + skip * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
gaps = []
end = rows[0][1]
last = rows[0][2]
for s, e, n in rows[1:]:
    if s > end:
        gaps.append((s - end, last, n, (s - rows[0][0]) / 1e3))
    if e > end:
        end, last = e, n
span = rows[-1][1] - rows[0][0]
print("window %.2f ms, idle %.2f ms in %d gaps" % (span / 1e6, sum(g[0] for g in gaps) / 1e6, len(gaps)))
for g in sorted(gaps, reverse=True)[:top]:
    print("%9.1f us idle at t=%9.1f us  after %-48s before %s" % (g[0] / 1e3, g[3], g[1], g[2]))
