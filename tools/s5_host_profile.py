#!/usr/bin/env python3
"""Where the host's CPU time of a configs[4] batch goes (BASELINE.json configs[4]: -i haplotypes -y 2 --use-hap-gibbs): one resident
batch, estimateBatch a few times ONE AT A TIME, every phase of the host layer with its wall time and the CPU time the whole process
spent meanwhile (RPVG_AMD_TRACE=1 RPVG_AMD_TRACE_CPU=1: teams of OpenMP threads included).

    OMP_WAIT_POLICY=passive RPVG_AMD_SINGLE_LANE=1 RPVG_AMD_TRACE=1 RPVG_AMD_TRACE_CPU=1 python tools/s5_host_profile.py [steps]

(one host lane: the phases do not overlap each other; the library's trace lines are collected from stderr and printed as one table)
"""
import collections
import os
import re
import resource
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpvg_amd import engine as eng_mod, synth  # noqa: E402
from rpvg_amd.batch import make_params  # noqa: E402


def thread_cpu():
    import glob
    ticks = os.sysconf("SC_CLK_TCK")
    out = {}
    for stat in glob.glob("/proc/self/task/*/stat"):
        try:
            text = open(stat).read()
            fields = text[text.rindex(")") + 2:].split()
            out[stat.split("/")[4]] = (int(fields[11]) / ticks, int(fields[12]) / ticks, text[text.index("(") + 1:text.rindex(")")])
        except Exception:  # noqa: BLE001
            pass
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    batch = synth.generate(seed=5, num_clusters=5000, total_paths=500000, total_reads=10000000, max_cluster_paths=5000)
    params = make_params(use_hap_gibbs=1)
    eng = eng_mod.Engine(0)
    prep = eng.prepare(batch)
    for _ in range(2):
        eng.run_raw("haplotypes", params, prep)
    tracing = bool(os.environ.get("RPVG_AMD_TRACE"))
    if tracing:  # (the library writes its trace to the process's stderr: into a file for the timed calls)
        sys.stderr.flush()
        saved, capture = os.dup(2), tempfile.TemporaryFile(mode="w+b")
        os.dup2(capture.fileno(), 2)
    th0 = thread_cpu()
    r0, t0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
    for _ in range(steps):
        eng.run_raw("haplotypes", params, prep)
    r1, t1 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
    cpu = (r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)
    if tracing:
        os.dup2(saved, 2)
        capture.seek(0)
        tot, cnt = collections.defaultdict(float), collections.Counter()
        for line in capture.read().decode(errors="replace").splitlines():
            m = re.match(r"\[rpvg_amd trace\] (.*?)\s+([\d.]+) ms$", line.rstrip())
            if m:
                tot[m.group(1).strip()] += float(m.group(2))
                cnt[m.group(1).strip()] += 1
            m = re.match(r"\[rpvg_hip trace\]\s+(.*?)\s+([\d.]+) ms$", line.rstrip())
            if m:
                tot["  library: " + m.group(1).strip()] += float(m.group(2))
                cnt["  library: " + m.group(1).strip()] += 1
        names = [k for k in tot if not k.endswith("[process CPU]")]
        print("phase                                                                      ms per call: wall   process CPU meanwhile")
        for k in sorted(names, key=lambda x: -tot.get(x + " [process CPU]", 0.0)):
            c = tot.get(k + " [process CPU]")
            print(f"{k:72s} {tot[k] / steps:8.2f}  " + (f"{c / steps:8.2f}" if c is not None else "       -"))
    print(f"{steps} calls: {1e3 * (t1 - t0) / steps:.2f} ms of wall time and {1e3 * cpu / steps:.2f} ms of CPU time per call")
    th1 = thread_cpu()
    rows = sorted(((th1[t][0] - th0.get(t, (0, 0))[0], th1[t][1] - th0.get(t, (0, 0))[1], t, th1[t][2]) for t in th1), key=lambda r: -(r[0] + r[1]))
    print(f"threads {len(rows)}; the busiest, ms of CPU time per call (user, system, tid, name; this thread is {os.getpid()}):")
    for r in rows[:12]:
        print(f"   {1e3 * r[0] / steps:7.2f} {1e3 * r[1] / steps:7.2f}  {r[2]} {r[3]}")
    prep.free()
    eng.close()


if __name__ == "__main__":
    main()
