#!/usr/bin/env python3
"""Share of wall time during which at least one kernel ran, from a rocprofv3 --kernel-trace csv directory of the pipelined bench
(several batches in flight: the per-step view of tools/gpu_gaps.py does not apply).  The window is the timed region's: from the
start of the (K + 1)-th last pairTile2Kernel launch to the end of the last kernel, K = the steps of the run's last loop (default:
the last 12 searches).  Copies run on the SDMA engines and are not kernels: they are not in the trace.

    python tools/gpu_busy_union.py <dir> [searches]
"""
import csv, glob, sys
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))[-1]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
ev.sort()
searches = [e for e in ev if e[2].startswith("pairTile2")]
t0 = searches[-last][0] if len(searches) >= last else ev[0][0]
t1 = max(e[1] for e in ev)
ks = [e for e in ev if e[1] > t0]
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
