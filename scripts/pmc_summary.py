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
        # the launches of one kernel fall into two kinds on this bench -- passes that search the whole cloud and passes that
        # certify nearly all of it -- told apart by the counter itself (above / below the middle of its range)
        mid = 0.5 * (min(vals) + max(vals))
        hi = [x for x in vals if x > mid] or vals
        lo = [x for x in vals if x <= mid] or vals
        print("%-34s n=%3d mean=%.6g first=%.6g last%d_mean=%.6g heavy_mean=%.6g (n=%d) light_mean=%.6g (n=%d)"
              % (k, len(vals), sum(vals) / len(vals), vals[0], len(t), sum(t) / len(t), sum(hi) / len(hi), len(hi), sum(lo) / len(lo), len(lo)))
