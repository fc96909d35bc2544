#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv) of a batch run: the durations of the pass kernels of ONE worker thread in launch
order, pass by pass (mean over that thread's gangs).  usage: pass_sequence.py <kernel_trace.csv> <passes per gang>"""
import csv
import sys
from collections import defaultdict

per = int(sys.argv[2])
rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "k_icp_cert_batch" in n or "k_icp_search_coop_batch" in n:
            rows[r["Thread_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "cert" if "cert" in n else "search"))
for tid, rs in sorted(rows.items()):
    rs.sort()
    for kind in ("cert", "search"):
        d = [(e - s) / 1e3 for s, e, k in rs if k == kind]
        gangs = len(d) // per
        if gangs == 0:
            continue
        mean = [sum(d[g * per + p] for g in range(gangs)) / gangs for p in range(per)]
        print("thread %s %-6s gangs %3d  per pass (us): %s" % (tid, kind, gangs, " ".join("%.0f" % v for v in mean)))
