#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel over one or more pass directories:

  python tools/pmc_kernels.py <dir> [<dir> ...] --kernel Search,pairTable

Prints, per kernel (name shortened), dispatches and every counter's total."""
import argparse
import collections
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", default="")
    args = ap.parse_args()
    keys = [k for k in args.kernel.split(",") if k]
    totals = collections.defaultdict(lambda: collections.defaultdict(float))
    dispatches = collections.defaultdict(set)
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "").replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                    if keys and not any(k in name for k in keys):
                        continue
                    totals[name][row["Counter_Name"]] += float(row["Counter_Value"])
                    dispatches[name].add((d, row.get("Dispatch_Id")))
    for name in sorted(totals):
        counters = totals[name]
        print(name)
        print(f"    {'DISPATCHES_PER_PASS':32s} {len(dispatches[name]) / max(1, len(args.dirs)):.6g}")
        for c in sorted(counters):
            print(f"    {c:32s} {counters[c]:.6g}")


if __name__ == "__main__":
    main()
