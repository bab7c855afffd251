#!/usr/bin/env python3
"""Share of wall time during which at least one kernel ran, from a rocprofv3 --kernel-trace csv directory of the pipelined bench
(several batches in flight: the per-step view of tools/gpu_gaps.py does not apply).  The window is the pipeline's steady state:
the K consecutive pairTile2Kernel launches (= batches) that lie closest together, from the start of the first to the start of the
launch behind them (default K = 12).  Copies run on the SDMA engines and are not kernels: they are not in the trace.

    python tools/gpu_busy_union.py <dir> [searches]
"""
import csv, glob, sys
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))[-1]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id")))
ev.sort()
# (the run's first engine — resident batches, two host lanes, two searches per batch — searches on its two main streams: not the pipeline's)
first_engine = []
for e in ev:
    if e[2].startswith("pairTile2") and e[3] not in first_engine:
        first_engine.append(e[3])
    if len(first_engine) == 2:
        break
searches = [e for e in ev if e[2].startswith("pairTile2") and e[3] not in first_engine]
ev = [e[:3] for e in ev]
# the pipeline's steady state: the `last` consecutive searches that lie closest together (the run also holds resident batches one at
# a time, warm-ups and a decoded run); the window runs from the start of the first of them to the start of the one behind them
best = None
for i in range(0, len(searches) - last):
    span = searches[i + last][0] - searches[i][0]
    if best is None or span < best[0]:
        best = (span, i)
if best is None:
    raise SystemExit("fewer searches in the trace than asked for")
t0, t1 = searches[best[1]][0], searches[best[1] + last][0]
ks = [(s, min(e, t1), n) for s, e, n in ev if e > t0 and s < t1]
busy, cur_s, cur_e = 0, None, None
depth_time = {}
for s, e, n in ks:
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
# concurrency: time-weighted mean number of kernels running
points = sorted([(max(s, t0), 1) for s, e, n in ks] + [(e, -1) for s, e, n in ks])
depth, prev, area = 0, t0, 0
for t, dlt in points:
    area += depth * (t - prev)
    prev, depth = t, depth + dlt
wall = t1 - t0
print(f"window {wall / 1e6:.2f} ms ({last} searches = batches), {len(ks)} kernels; at least one kernel running {busy / 1e6:.2f} ms = {busy / wall:.3f} of it; "
      f"mean kernels running {area / wall:.2f}; per batch {wall / 1e6 / last:.2f} ms")
by = {}
for s, e, n in ks:
    by[n] = by.get(n, 0) + (e - max(s, t0))
for n, t in sorted(by.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {n[:60]:60s} {t / 1e6 / last:8.3f} ms of kernel time per batch")
