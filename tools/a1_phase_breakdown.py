#!/usr/bin/env python3
"""Phase breakdown of the drop-in path (PathEstimator::estimate() once per cluster from an OpenMP team, bench.py --workload a1)
from the host layer's own traces:

    RPVG_AMD_TRACE=1    python bench.py --workload a1 ...  2> phase_totals.txt     (per pass: seconds summed over threads)
    RPVG_AMD_TIMELINE=1 python bench.py --workload a1 ...  2> host_timeline.txt    (every phase of every batch with its thread)

    python tools/a1_phase_breakdown.py phase_totals.txt host_timeline.txt [team]

Prints, per pass of 5 000 calls: the phase totals, and from the timeline the batches (number, mean clusters, wall time of a
batch by phase), how many batches are on the GPU at once, and the split of the pass into its head (the large clusters the
reference's descending order hands out first: batches of several milliseconds, all device contexts busy) and its tail (rounds
of the whole team in one batch)."""
import os
import re
import sys


def totals(path):
    passes, cur = [], {}
    for line in open(path, errors="replace"):
        m = re.match(r"\[rpvg_amd trace\] (.*?)\s+([\d.]+) ms$", line.rstrip())
        if not m:
            continue
        name, ms = m.group(1).strip(), float(m.group(2))
        if name in cur:  # the next report
            passes.append(cur)
            cur = {}
        cur[name] = ms
    if cur:
        passes.append(cur)
    return [p for p in passes if "combiner: number of batches" in p]


def timeline(path):
    rows = []
    for line in open(path, errors="replace"):
        m = re.match(r"\[timeline\] thread (\d+) (.*?)\s+([\d.]+)\s+([\d.]+)$", line.rstrip())
        if m:
            rows.append((int(m.group(1)), m.group(2).strip(), float(m.group(3)), float(m.group(4))))
    return rows


def main():
    tot = totals(sys.argv[1])
    team = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    print(f"== phase totals per pass (ms summed over the threads that ran the phase; team of {team}) ==")
    keys = ["combiner: flatten the cluster (callers, summed)", "combiner: batches (leaders, summed)", "combiner: join the clusters", "combiner: upload",
            "combiner: containers in", "combiner: estimateBatch", "combiner: containers out", "nested: matrices + search + subsets + EM + merge on the device",
            "nested: search + subsets + EM on the device", "nested: estimates out of the device block"]
    for i, p in enumerate(tot):
        batches = p.get("combiner: number of batches", 0.0)
        print(f"pass {i}: {batches:.0f} batches, {p.get('combiner: clusters in batches', 0.0) / max(batches, 1):.1f} clusters per batch")
        for k in keys:
            if k in p:
                print(f"    {k:68s} {p[k]:9.1f} ms   {p[k] / max(batches, 1):7.3f} ms per batch")
    if len(sys.argv) < 3:
        return
    rows = timeline(sys.argv[2])
    batches = sorted((r for r in rows if r[1].startswith("combiner: batches")), key=lambda r: r[2])
    if not batches:
        return
    # passes: the harness' own phase around its loop of calls (runner_capi.cpp, rpvg_amd_run_team)
    passes = sorted((r for r in rows if r[1].startswith("team: one pass")), key=lambda r: r[2])
    if not passes and os.environ.get("A1_PASS_BOUNDS"):  # (a trace without the marker — round 5's library: the boundaries read off the batches' gaps)
        bounds = [float(x) for x in os.environ["A1_PASS_BOUNDS"].split(",")]
        passes = [(0, "team: one pass", a, b) for a, b in zip(bounds[:-1], bounds[1:])]
    spans = [[b for b in batches if p[2] - 1e-3 <= b[2] and b[3] <= p[3] + 1e-3] for p in passes] if passes else [batches]
    spans = [sp for sp in spans if sp]

    def child(thread, s, e, name):
        for r in rows:
            if r[0] == thread and r[1].startswith(name) and r[2] >= s - 1e-3 and r[3] <= e + 1e-3:
                return r[3] - r[2]
        return 0.0

    print("\n== batches of every pass from the timeline (the traced run itself is slower than the line: it prints 15 000 events) ==")
    for i, sp in enumerate(spans):
        t0, t1 = (passes[i][2], passes[i][3]) if passes else (sp[0][2], max(x[3] for x in sp))
        wall = t1 - t0
        busy = sum(x[3] - x[2] for x in sp)
        # head / tail: the tail starts at the first batch after which never more than one batch is on the GPU for 10 batches running
        ends = sorted(x[3] for x in sp)
        tail_from = None
        for j in range(len(sp)):
            window = sp[j:j + 10]
            if len(window) == 10 and all(sum(1 for y in sp if y[2] < x[2] < y[3]) == 0 for x in window):
                tail_from = sp[j][2]
                break
        head = [x for x in sp if tail_from is None or x[2] < tail_from]
        tail = [x for x in sp if tail_from is not None and x[2] >= tail_from]

        def part(name, xs):
            if not xs:
                return
            w = max(x[3] for x in xs) - xs[0][2]
            up = sum(child(x[0], x[2], x[3], "combiner: upload") for x in xs) / len(xs)
            est = sum(child(x[0], x[2], x[3], "combiner: estimateBatch") for x in xs) / len(xs)
            join = sum(child(x[0], x[2], x[3], "combiner: join the clusters") for x in xs) / len(xs)
            print(f"    {name}: {w:7.1f} ms wall, {len(xs):4d} batches, mean batch {sum(x[3] - x[2] for x in xs) / len(xs):6.2f} ms "
                  f"(join {join:.2f}, upload {up:.2f}, estimateBatch {est:.2f}), batches on the GPU at once {sum(x[3] - x[2] for x in xs) / max(w, 1e-9):.2f}, "
                  f"a batch starts every {w / len(xs):.2f} ms")
        print(f"pass {i}: {wall:7.1f} ms wall, {len(sp)} batches, leaders busy {busy:.1f} ms")
        part("head (large clusters)", head)
        part("tail (the team in step)", tail)
        _ = ends


if __name__ == "__main__":
    main()
