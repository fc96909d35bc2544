#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: counter value per dispatch (in dispatch order) and the mean over the
last `tail` dispatches (the steady-state passes).  usage: pmc_summary.py <dir> [kernel substring] [tail]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_icp_search_coop"
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 4
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if want in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    for k, v in sorted(acc.items()):
        v.sort()
        vals = [x for _, x in v]
        t = vals[-tail:]
        print("%-34s n=%3d mean=%.6g first=%.6g last%d_mean=%.6g" % (k, len(vals), sum(vals) / len(vals), vals[0], len(t), sum(t) / len(t)))
