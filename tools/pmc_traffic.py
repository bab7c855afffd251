#!/usr/bin/env python3
"""Folds two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as the TCC counter slots
require: MI355X_MICROARCH.md, counters table) into the HBM traffic of one kernel family.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <fetch_dir> -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <write_dir> -- python bench.py ...
  python tools/pmc_traffic.py --fetch-dir <fetch_dir> --write-dir <write_dir> --kernel emSparseKernel \
         --steps 2 --out profiles/pmc_traffic_s3.json [--double-fetch]

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
    if args.shape:
        rows, cols, ld = (int(x) for x in args.shape.split(","))
        out["shape"] = dict(rows=rows, cols=cols, ld=ld)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
