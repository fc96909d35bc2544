#!/usr/bin/env python3
"""Print (calls, average, min, max in microseconds) of the kernels whose name contains any of the given substrings, from the
kernel_stats.csv of a rocprofv3 --kernel-trace --stats run.  usage: kstats.py <dir> [substring ...]"""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if len(sys.argv) < 3 or any(w in n for w in sys.argv[2:]):
            short = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
            if n.startswith("void (anonymous") or n.startswith("(anonymous"):
                short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            print("%-44s calls %4s avg %10.1f min %10.1f max %10.1f us" % (short[:44], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
