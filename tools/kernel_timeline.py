#!/usr/bin/env python3
"""Start/end (us) of the kernels of the last hot-path step in a rocprofv3 --kernel-trace csv directory (kernels >= min_us)."""
import csv, glob, sys
d, min_us = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))[-1]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id"), r.get("Stream_Id")))
ev.sort()
idx = [i for i, e in enumerate(ev) if e[2].startswith("partitionRows")]
start = ev[idx[-2]][0]
for s, e, n, q, st in ev:
    if s >= start - 1000 and (e - s) / 1e3 >= min_us and "Dense" not in n:
        print(f"{(s - start) / 1e3:9.1f} {(e - start) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q} s{st} {n[:50]}")
