#!/usr/bin/env python3
"""Interleaved A/B of bench.py variants on one box: every variant (a set of environment variables) is run `rounds` times in
turn, and the medians of the single-step times are compared — box-to-box and run-to-run spread of the default 20-step mean is
+-0.5 ms, more than most changes are worth.

    python tools/ab_bench.py [--rounds 3] [--steps 60] [--workload s3] "NAME=ENV1=x ENV2=y" "base=" ...
"""
import argparse
import json
import os
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--workload", default="s3")
ap.add_argument("variants", nargs="+")
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
results = {}
for rnd in range(args.rounds):
    for v in args.variants:
        name, _, envs = v.partition("=")
        env = dict(os.environ)
        for kv in envs.split():
            k, _, val = kv.partition("=")
            env[k] = val
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", args.workload, "--steps", str(args.steps), "--warmup", "5",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:  # noqa: BLE001
            print(name, "FAILED", out.stderr[-400:])
            continue
        sp = d.get("ms_per_step_spread", {})
        results.setdefault(name, []).append((d["ms_per_step"], sp.get("median"), sp.get("min"), d.get("ms_per_step_resident"), d.get("host_cpu_ms_per_step")))
for name, rs in results.items():
    fmt = lambda i: " ".join(f"{r[i]:.2f}" if r[i] is not None else "-" for r in rs)
    print(f"{name:28s} mean [{fmt(0)}]  median [{fmt(1)}]  min [{fmt(2)}]  resident [{fmt(3)}]  host cpu [{fmt(4)}]")
