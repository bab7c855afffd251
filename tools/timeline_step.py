#!/usr/bin/env python3
"""Prints the phases of one step of a RPVG_AMD_TIMELINE=1 run: stderr log, then the step counted from the end
(default 1 = the last one; bench.py's last call is the decoded run after the timed steps, so 3 is a timed step)."""
import sys
ev = []
for line in open(sys.argv[1]):
    if not line.startswith("[timeline]"):
        continue
    r = line.split()
    try:
        ev.append((float(r[-2]), float(r[-1]), int(r[2]), " ".join(r[3:-2])))
    except ValueError:
        pass
ev.sort()
# last step = from the last "estimateBatch incl. teardown" start
starts = [e for e in ev if "estimateBatch" in e[3]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = starts[-back][0]
t1 = starts[-back + 1][0] if back > 1 else float("inf")
for s, e, t, n in ev:
    if t0 - 0.01 <= s < t1 - 0.01:
        print(f"{s - t0:7.2f} {e - t0:7.2f} {e - s:6.2f} T{t} {n}")
