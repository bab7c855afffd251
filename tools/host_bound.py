#!/usr/bin/env python3
"""Predicted host bound of the 8-GPU curve from ONE GPU: the default workload with the host budget one rank
of an N-rank node has — the thread team hostThreads() gives a rank when LOCAL_WORLD_SIZE = N, confined (taskset) to the
rank's share of the CPUs the cgroup grants the node (cpu.max; all hardware threads if it grants everything).  The step time
under that budget is what a rank of the node can reach at best if the ranks do not otherwise get in each other's way; against
the unconstrained step it is the weak-scaling efficiency the host alone allows.

    python tools/host_bound.py [--ranks 8] [--steps 40]
"""
import argparse
import json
import os
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--workload", default="s3")
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quota_cpus():
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return float(quota) / float(period)
    except OSError:
        pass
    return float(os.cpu_count() or 1)


def run(env_extra, prefix):
    env = dict(os.environ, **env_extra)
    cmd = prefix + [sys.executable, os.path.join(root, "bench.py"), "--workload", args.workload, "--steps", str(args.steps), "--warmup", "5", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    sp = d.get("ms_per_step_spread", {})
    return dict(ms_per_step=d["ms_per_step"], median=sp.get("median"), host_cpu_ms_per_step=d.get("host_cpu_ms_per_step"), host_threads_per_lane=d.get("host_threads_per_lane"))


cpus = quota_cpus()
share = max(1, int(cpus // args.ranks))
free = run({}, [])
bound = run({"LOCAL_WORLD_SIZE": str(args.ranks)}, ["taskset", "-c", f"0-{share - 1}"])
line = dict(ranks=args.ranks, node_cpus=cpus, cpus_per_rank=share, unconstrained=free, rank_budget=bound,
            predicted_host_bound_efficiency=(free["median"] or free["ms_per_step"]) / (bound["median"] or bound["ms_per_step"]),
            note="rank_budget = the same bench with LOCAL_WORLD_SIZE=ranks (the rank's thread team) under taskset to the rank's share of the "
                 "node's CPUs; efficiency = unconstrained step / budgeted step: what the host side alone allows an N-rank node")
print(json.dumps(line))
