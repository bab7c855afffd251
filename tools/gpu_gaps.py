#!/usr/bin/env python3
"""GPU idle gaps inside the hot-path steps of a rocprofv3 --kernel-trace csv directory: for every step (from one
partitionRowsKernel pair to the next), the union of the kernel intervals and the gaps of at least min_us between them."""
import csv, glob, sys
d, min_us = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))[-1]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
ev.sort()
starts = [e[0] for e in ev if e[2].startswith("partitionRows")][0::2]
for si in range(len(starts) - 4, len(starts) - 1):
    t0, t1 = starts[si], starts[si + 1]
    ks = [e for e in ev if t0 <= e[0] < t1]
    busy, cur_s, cur_e, gaps = 0, ks[0][0], ks[0][1], []
    last_name = ks[0][2]
    for s, e, n in ks[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            if (s - cur_e) / 1e3 >= min_us:
                gaps.append(((cur_e - t0) / 1e3, (s - cur_e) / 1e3, last_name, n))
            cur_s, cur_e = s, e
            last_name = n
        elif e > cur_e:
            cur_e = e
            last_name = n
    busy += cur_e - cur_s
    print(f"step {si}: {(t1 - t0) / 1e3:.0f} us, GPU busy {busy / 1e3:.0f} us, last kernel ends at {(cur_e - t0) / 1e3:.0f} us")
    for at, length, before, after in gaps:
        print(f"    idle {length:6.0f} us at {at:7.0f}: after {before[:34]} before {after[:34]}")
