#!/usr/bin/env python3
"""Prints the average duration of the kernels matching the given substrings from a rocprofv3 --kernel-trace --stats run."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + "/*/*kernel_stats.csv"))[-1]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if any(k in n for k in sys.argv[2:]):
        out.append(f"{n[-36:]}={float(r['AverageNs'])/1e3:.0f}us")
print(" ".join(out))
