#!/usr/bin/env python3
"""Folds two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as the TCC counter slots
require: MI355X_MICROARCH.md, counters table) into the HBM traffic of one kernel family.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <fetch_dir> -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <write_dir> -- python bench.py ...
  python tools/pmc_traffic.py --fetch-dir <fetch_dir> --write-dir <write_dir> --kernel emSparseKernel \
         --steps 2 --out profiles/r03/pmc_traffic_s3.json [--double-fetch]

--double-fetch applies the guide's gfx950 correction for wide (16 B/lane) streaming reads; other access widths are
uncalibrated and left as reported.  FETCH_SIZE / WRITE_SIZE are in KB."""
import argparse
import csv
import glob
import json
import os


def counter_rows(directory):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                yield row


def total(directory, counter, kernel_substr):
    value, dispatches = 0.0, set()
    for row in counter_rows(directory):
        if row.get("Counter_Name") != counter or not any(k in row.get("Kernel_Name", "") for k in kernel_substr.split(",")):
            continue
        value += float(row["Counter_Value"])
        dispatches.add(row.get("Dispatch_Id"))
    return value, len(dispatches)


def short_name(kernel_name):
    """`void (anonymous namespace)::emRegisterKernel<1, 16>(EmLaunchArgs)` -> `emRegisterKernel<1,16>`"""
    n = kernel_name.replace("(anonymous namespace)::", "").replace("void ", "")
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return n[:cut].replace(" ", "")


def per_kernel(directory, counter, kernel_substr):
    """counter total and dispatch count per kernel name"""
    out = {}
    for row in counter_rows(directory):
        if row.get("Counter_Name") != counter or not any(k in row.get("Kernel_Name", "") for k in kernel_substr.split(",")):
            continue
        rec = out.setdefault(short_name(row["Kernel_Name"]), [0.0, set()])
        rec[0] += float(row["Counter_Value"])
        rec[1].add(row.get("Dispatch_Id"))
    return {k: (v[0], len(v[1])) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch-dir", required=True)
    ap.add_argument("--write-dir", required=True)
    ap.add_argument("--kernel", required=True, help="substring(s) of the kernel name, comma separated")
    ap.add_argument("--steps", type=int, required=True, help="hot-path passes (warmup + timed) the profiled command ran")
    ap.add_argument("--double-fetch", action="store_true")
    ap.add_argument("--command", default="")
    ap.add_argument("--shape", default="", help="rows,cols,ld of the matrix the kernel streamed (recorded; bench.py checks it)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--commit", default="", help="commit of the tree the passes ran on (recorded)")
    args = ap.parse_args()
    fetch_kb, n_fetch = total(args.fetch_dir, "FETCH_SIZE", args.kernel)
    write_kb, n_write = total(args.write_dir, "WRITE_SIZE", args.kernel)
    fetch_bytes = fetch_kb * 1024.0 * (2.0 if args.double_fetch else 1.0)
    write_bytes = write_kb * 1024.0
    out = dict(command=args.command, kernel=args.kernel, steps=args.steps, dispatches_fetch_pass=n_fetch, dispatches_write_pass=n_write,
               FETCH_SIZE_KB_total=fetch_kb, WRITE_SIZE_KB_total=write_kb,
               correction=("FETCH_SIZE doubled (gfx950, wide 16 B/lane streaming reads)" if args.double_fetch else
                           "none: the kernel's loads are 4 and 8 B per lane, for which the guide gives no calibration"),
               traffic_bytes_per_step=(fetch_bytes + write_bytes) / args.steps,
               fetch_bytes_per_step=fetch_bytes / args.steps, write_bytes_per_step=write_bytes / args.steps)
    # the same per kernel variant and launch (dispatch)
    fetch_k, write_k = per_kernel(args.fetch_dir, "FETCH_SIZE", args.kernel), per_kernel(args.write_dir, "WRITE_SIZE", args.kernel)
    out["per_kernel"] = {}
    for name in sorted(set(fetch_k) | set(write_k)):
        f_kb, f_n = fetch_k.get(name, (0.0, 0))
        w_kb, w_n = write_k.get(name, (0.0, 0))
        fb = f_kb * 1024.0 * (2.0 if args.double_fetch else 1.0)
        out["per_kernel"][name] = dict(launches_fetch_pass=f_n, launches_write_pass=w_n,
                                       traffic_bytes_per_launch=(fb / max(1, f_n)) + (w_kb * 1024.0 / max(1, w_n)))
    out["commit"] = args.commit
    out["source"] = "profiles/%s (tools/pmc_traffic.py)" % os.path.basename(args.out)
    if args.shape:
        rows, cols, ld = (int(x) for x in args.shape.split(","))
        out["shape"] = dict(rows=rows, cols=cols, ld=ld)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
