#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_icp_pass"
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if want in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-40s n=%3d mean=%.6g" % (k, len(v), sum(v) / len(v)))
