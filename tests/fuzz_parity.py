#!/usr/bin/env python3
"""Randomised parity sweep: GPU engine vs the CPU oracle over seeds, batch shapes, models and option values.
The open-ended sweep is run by hand (minutes of GPU time; prints every mismatch and a summary); a fixed set of its
seeds is collected under -m gpu (tests/test_hip_collapse.py::test_trimmed_parity_sweep).

  python tests/fuzz_parity.py [rounds] [first_seed] [gibbs]
(gibbs: every case is a haplotype model with --use-hap-gibbs — the sampler's chains on the device for ploidy 1 and 2,
the host-driven sampler for ploidy 3)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import pyoracle  # noqa: E402
from rpvg_amd import engine as eng_mod, synth  # noqa: E402
from rpvg_amd.batch import ClusterBatch, make_params  # noqa: E402
from tests import small_cases  # noqa: E402

REL = 1e-6


def compare(got, ref, check_iters=True):
    problems = []
    if len(got) != len(ref):
        return [f"cluster count {len(got)} vs {len(ref)}"]
    for k, (g, r) in enumerate(zip(got, ref)):
        gk, rk = g.keyed(), r.keyed()
        if set(gk) != set(rk):
            # Two columns whose marginal posteriors are equal up to rounding are visited in an order that the rounding
            # decides, on either side (seen: seed 5109, a 12-row cluster, pair (1, 2) vs (2, 1)): the same diplotype
            # written in the other order is not a mismatch.
            gk = {tuple(sorted(key)): v for key, v in gk.items()}
            rk = {tuple(sorted(key)): v for key, v in rk.items()}
        if set(gk) != set(rk):
            # ... and the tie decides which of the two columns is visited first, whose pair with itself is then kept with
            # posterior exactly 0 (a "late loser" of the search: seed 21204, an 11-row cluster, (0, 0) on one side and
            # (1, 1) on the other next to the same diplotype (0, 1) with posterior 1): sets of weight zero carry nothing.
            gk = {key: v for key, v in gk.items() if v[0] != 0.0 or len(v[1])}
            rk = {key: v for key, v in rk.items() if v[0] != 0.0 or len(v[1])}
        if set(gk) != set(rk):
            problems.append(f"cluster {k}: group sets differ ({len(gk)} vs {len(rk)})")
            continue
        for key, (post, ab) in rk.items():
            if not small_cases.rel_close(gk[key][0], post, rel=REL, floor=1e-8):
                problems.append(f"cluster {k} set {key}: posterior {gk[key][0]} vs {post}")
            if not small_cases.rel_close(gk[key][1], ab, rel=REL):
                problems.append(f"cluster {k} set {key}: abundance {gk[key][1]} vs {ab}")
        if g.total_count != r.total_count:
            problems.append(f"cluster {k}: total {g.total_count} vs {r.total_count}")
        if abs(g.noise_count - r.noise_count) > REL * max(1.0, r.total_count):
            problems.append(f"cluster {k}: noise {g.noise_count} vs {r.noise_count}")
        if check_iters and dict(zip(g.em_cols, g.em_iters)) != dict(zip(r.em_cols, r.em_iters)):
            problems.append(f"cluster {k}: EM iterations differ")
    return problems


def draw_case(seed, only_gibbs=False):
    """Batch, model and options of one seed of the sweep."""
    rng = np.random.default_rng(seed)
    shape = rng.integers(0, 3)
    if shape == 0:      # hand-built small clusters incl. empty ones
        batch = ClusterBatch.from_clusters(small_cases.make_batch_clusters(seed, n_clusters=int(rng.integers(1, 200)),
                                                                           max_reads=int(rng.integers(25, 400))))
    elif shape == 1:    # generator: many small clusters
        batch = synth.generate(seed=seed, num_clusters=int(rng.integers(70, 400)), total_paths=int(rng.integers(2000, 12000)),
                               total_reads=int(rng.integers(20000, 400000)))
    else:               # generator: few large clusters
        batch = synth.generate(seed=seed, num_clusters=int(rng.integers(3, 30)), total_paths=int(rng.integers(1500, 8000)),
                               total_reads=int(rng.integers(50000, 600000)), max_cluster_paths=int(rng.integers(200, 4000)))
    model = ["transcripts", "haplotype-transcripts", "haplotypes", "strains"][int(rng.integers(0, 4))]
    rows = np.diff(batch.cluster_row_off.astype(np.float64))
    paths = np.diff(batch.cluster_path_off.astype(np.float64))
    if model in ("strains", "haplotypes") and float((rows * paths * paths).sum()) > 2e9:
        model = "transcripts"  # the oracle's pair enumeration / greedy cover is O(rows x paths^2) per cluster
    kw = dict(max_em_its=int(rng.choice([3, 50, 10000])), max_rel_em_conv=float(rng.choice([1e-3, 1e-2, 1e-5])),
              min_hap_prob=float(rng.choice([1e-3, 1e-2, 1e-5])), rng_seed=int(rng.integers(0, 1000)))
    if only_gibbs and model not in ("haplotype-transcripts", "haplotypes"):
        model = "haplotypes" if model == "strains" else "haplotype-transcripts"
    if model in ("haplotype-transcripts", "haplotypes"):
        kw["ploidy"] = int(rng.choice([1, 2, 2, 2, 3]))
        kw["use_hap_gibbs"] = int(only_gibbs or rng.random() < 0.25)
    if model == "haplotype-transcripts" and rng.random() < 0.2 and kw["min_hap_prob"] >= 1e-3:
        # (with 1e-5 the reference's own assertion sum_hap_prob <= 1 fails on the rounding of 100 000 weights,
        # src/path_abundance_estimator.cpp:748)
        kw["ind_hap_inference"] = 1
    if model == "haplotype-transcripts" and kw.get("ploidy") == 3 and kw["min_hap_prob"] < 1e-3:
        # (seed 90144: tens of thousands of triplets above 1e-5 — the reference's own assertion sum_hap_prob <= 1,
        # src/path_abundance_estimator.cpp:748, fails on the rounding of their weights, in the oracle as in the reference)
        kw["min_hap_prob"] = 1e-3
    if model == "haplotypes" and kw["ploidy"] == 3 and (batch.num_paths > 3000 or float((rows * paths ** 3).sum()) > 6e10):
        kw["ploidy"] = 2  # full enumeration of triplets over thousands of paths is not a test case (seed 50091: 465 paths in one
        # cluster, 16.7 M triplets over its rows: the oracle alone ran into the sweep's 15-minute limit)
    return dict(seed=seed, shape=int(shape), batch=batch, model=model, kw=kw)


def oracle_in_a_child(model, kw, batch, oracle_threads):
    """The oracle's estimates from a process of its own (tests/oracle_child.py); None: it died — the restatement keeps the
    reference's assertions, and one of them fails on rounding for thresholds far from the defaults."""
    import pickle
    import subprocess
    child = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_child.py")],
                           input=pickle.dumps((model, kw, batch, oracle_threads), protocol=pickle.HIGHEST_PROTOCOL), capture_output=True)
    if child.returncode == 0 and child.stdout:
        return pickle.loads(child.stdout)
    stderr = child.stderr.decode(errors="replace")
    if "sum_hap_prob" in stderr and "Assertion" in stderr:  # the one abort that is the reference's own (src/path_abundance_estimator.cpp:748)
        return None
    raise RuntimeError(f"the oracle process failed (exit code {child.returncode}): {stderr[-600:]}")


def run_case(eng, case, oracle_threads=32, isolate_oracle=False):
    """Mismatches between the engine and the oracle on one case (empty list: parity)."""
    params = make_params(**case["kw"])
    if isolate_oracle:
        ref = oracle_in_a_child(case["model"], case["kw"], case["batch"], oracle_threads)
    else:
        ref, _ = pyoracle.run(case["model"], params, case["batch"], oracle_threads)
    team = int(os.environ.get("RPVG_FUZZ_TEAM", "0"))
    if team > 0:  # the drop-in path: PathEstimator::estimate() once per cluster from an OpenMP team (the call combiner, page-locked segments)
        got, _ = eng.run_team(case["model"], params, eng.prepare(case["batch"], per_cluster=True), team)
    else:
        got, _ = eng.run(case["model"], params, eng.prepare(case["batch"]))
    if ref is None:
        # nothing to compare with: the engine still has to run through, conserve the read mass and keep its posteriors in [0, 1]
        bad = [f"cluster {k}: mass not conserved" for k, g in enumerate(got)
               if len(g.abundances) and abs(float(np.sum(g.abundances)) + g.noise_count - g.total_count) > 1e-6 * max(1.0, g.total_count)]
        bad += [f"cluster {k}: posterior outside [0, 1]" for k, g in enumerate(got) if len(g.posteriors) and (g.posteriors.min() < 0 or g.posteriors.max() > 1 + 1e-9)]
        return bad or ["skipped: the oracle stopped at the reference's own assertion sum_hap_prob <= 1 (src/path_abundance_estimator.cpp:748)"]
    return compare(got, ref)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    only_gibbs = len(sys.argv) > 3 and sys.argv[3] == "gibbs"
    eng = eng_mod.Engine(0)
    failures = 0
    skipped = 0
    t0 = time.time()
    for i in range(rounds):
        seed = seed0 + i
        case = draw_case(seed, only_gibbs)
        try:
            problems = run_case(eng, case, isolate_oracle=True)
            if problems and problems[0].startswith("skipped"):
                print(f"{time.time() - t0:6.0f}s [{i:3d}] seed {seed} shape {case['shape']} {case['model']:22s} {case['kw']} "
                      f"clusters {case['batch'].num_clusters} -> {problems[0]}", flush=True)
                skipped += 1
                continue
        except Exception as exc:  # noqa: BLE001
            problems = [f"exception: {exc}"]
        status = "ok" if not problems else "MISMATCH"
        print(f"{time.time() - t0:6.0f}s [{i:3d}] seed {seed} shape {case['shape']} {case['model']:22s} {case['kw']} "
              f"clusters {case['batch'].num_clusters} -> {status}", flush=True)
        for p in problems[:5]:
            print("      ", p, flush=True)
        failures += bool(problems)
    print(f"{rounds} rounds, {failures} with mismatches, {skipped} without an oracle (the reference's own assertion), {time.time() - t0:.0f} s")
    eng.close()
    # (a sweep that mostly skips has compared nothing: one case in two hundred is the rate seen)
    sys.exit(1 if failures or skipped > max(2, rounds // 20) else 0)


if __name__ == "__main__":
    main()
