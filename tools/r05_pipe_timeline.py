#!/usr/bin/env python3
"""Kernels (>= min_us) of a stretch of the pipeline's steady state from a rocprofv3 --kernel-trace csv directory: start, end, duration,
queue, stream, name — who ran next to whom.  python tools/r05_pipe_timeline.py <dir> [window_ms] [min_us]"""
import csv, glob, sys
d = sys.argv[1]
window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))[-1]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id"), r.get("Stream_Id")))
ev.sort()
first_engine = []
for e in ev:
    if e[2].startswith("pairTile2") and e[4] not in first_engine:
        first_engine.append(e[4])
    if len(first_engine) == 2:
        break
searches = [e for e in ev if e[2].startswith("pairTile2") and e[4] not in first_engine]  # (the pipeline's workers, not the run's first engine)
last = 16
best = None
for i in range(0, len(searches) - last):
    span = searches[i + last][0] - searches[i][0]
    if best is None or span < best[0]:
        best = (span, i)
t0 = searches[best[1] + 4][0]
t1 = t0 + int(window_ms * 1e6)
for s, e, n, q, st in ev:
    if s >= t0 and s < t1 and (e - s) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q} s{st} {n[:48]}")
