#!/usr/bin/env python3
"""Average duration (ms) of every phase over the last N steps of a RPVG_AMD_TIMELINE=1 log, per host thread."""
import collections
import sys
log, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = []
for line in open(log):
    if not line.startswith("[timeline]"):
        continue
    r = line.split()
    try:
        ev.append((float(r[-2]), float(r[-1]), int(r[2]), " ".join(r[3:-2])))
    except ValueError:
        pass
ev.sort()
starts = [e[0] for e in ev if "estimateBatch" in e[3]][-last:]
acc = collections.defaultdict(list)
for s, e, t, n in ev:
    if s >= starts[0] - 0.01:
        acc[(t, n)].append(e - s)
for (t, n), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / len(starts) > 0.25:
        print(f"T{t} {n:48s} {sum(v) / len(starts):7.2f}")
