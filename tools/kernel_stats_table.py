#!/usr/bin/env python3
"""Prints a rocprofv3 kernel_stats.csv as a table (kernels above min_total_ms)."""
import csv, sys
path, min_ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in csv.DictReader(open(path)):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-52:]
    tot = float(r["TotalDurationNs"]) / 1e6
    if tot >= min_ms:
        print(f"{n:54s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us total {tot:8.1f} ms")
