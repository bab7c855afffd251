#!/usr/bin/env python3
"""Prints the phases of the last step of a RPVG_AMD_TIMELINE=1 run (stderr log given as argument)."""
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
t0 = starts[-1][0]
for s, e, t, n in ev:
    if s >= t0 - 0.01:
        print(f"{s - t0:7.2f} {e - t0:7.2f} {e - s:6.2f} T{t} {n}")
