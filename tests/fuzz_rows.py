#!/usr/bin/env python3
"""Randomised sweep of row construction (GPU vs the oracle): seeds, cluster shapes, precision / noise options,
single- and paired-end, name-group collapsing, reads wider than the 16-lane kernel.  The open-ended sweep is run by hand;
tests/test_hip_rows.py::test_row_merge_sweep_default_precision collects 1 000 of its configurations under -m gpu.

  python tests/fuzz_rows.py [rounds] [first_seed] [fixed prob_precision]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import pyoracle  # noqa: E402
from rpvg_amd import hip  # noqa: E402
from rpvg_amd.rows import AlignmentBatch, RowParams  # noqa: E402
from tests import test_hip_rows as T  # noqa: E402
from tests import test_row_construction as kat  # noqa: E402


def draw_case(seed, fixed_precision=None, allow_chains=True):
    rng = np.random.default_rng(seed)
    chains = bool(rng.random() < 0.3)
    collapse = bool(rng.random() < 0.25)
    wide = bool(rng.random() < 0.15)
    single_end = bool(rng.random() < 0.3)
    precision = float(rng.choice([1e-8, 1e-8, 1e-6, 1e-3, 0.05]))
    min_noise = float(rng.choice([0.0, 1e-4, 1e-2]))
    if fixed_precision is not None:
        precision = fixed_precision
    if not allow_chains:
        chains = False
    clusters = T.make_alignment_clusters(seed, n_clusters=int(rng.integers(2, 12)), max_paths=int(rng.integers(1, 120)),
                                         reads_per_cluster=int(rng.integers(1, 3000 if not wide else 80)), collapse=collapse,
                                         wide=wide, chains=chains)
    batch = AlignmentBatch.from_clusters(clusters)
    prm = RowParams(prob_precision=precision, min_noise_prob=min_noise, is_single_end=single_end,
                    frag_length_log_prob=None if single_end else kat.frag_table())
    return dict(seed=seed, chains=chains, collapse=collapse, wide=wide, single_end=single_end, precision=precision, min_noise=min_noise,
                batch=batch, params=prm)


def run_case(ctx, case):
    """'exact': unmerged and merged rows equal the oracle's; 'order-dependent': the merged rows are a valid merge of the
    same rows that differs from the oracle's where the reference's tolerant operator< is not transitive.  Raises on
    anything else."""
    batch, prm = case["batch"], case["params"]
    ref, _ = pyoracle.build_rows(batch, prm, merge=False)
    got, _, _ = ctx.build_rows(batch, prm, merge=False)
    T.compare_unmerged(got, ref)
    ref_m, _ = pyoracle.build_rows(batch, prm, merge=True)
    got_m, _, _ = ctx.build_rows(batch, prm, merge=True)
    try:
        T.compare_merged(got_m, ref_m)
        return "exact"
    except AssertionError:
        # Rows that the tolerant operator< calls equal but quickMergeIdentical does not: which of them end up next
        # to each other — and therefore how many merge — is the sort's business on either side (chains of
        # near-equal noise terms, coarse precisions).  What must hold: reads conserved per row structure, heads kept
        # bit for bit, row count close.
        T.check_valid_merge(got_m, got, ref_m, count_tolerance=0.5)
        return "order-dependent"


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    fixed_precision = float(sys.argv[3]) if len(sys.argv) > 3 else None
    ctx = hip.Context(0)
    failures = 0
    order_dependent = 0
    t0 = time.time()
    for i in range(rounds):
        case = draw_case(seed0 + i, fixed_precision)
        problems = []
        try:
            if run_case(ctx, case) == "order-dependent":
                order_dependent += 1
                print(f"      order-dependent merge: seed {case['seed']} chains={case['chains']}", flush=True)
        except AssertionError as exc:
            problems.append(f"assertion: {str(exc)[:300]}")
        except Exception as exc:  # noqa: BLE001
            problems.append(f"exception: {exc}")
        status = "ok" if not problems else "MISMATCH"
        print(f"{time.time() - t0:5.0f}s [{i:3d}] seed {case['seed']} chains={case['chains']} collapse={case['collapse']} wide={case['wide']} "
              f"single_end={case['single_end']} precision={case['precision']} min_noise={case['min_noise']} reads={case['batch'].num_reads} -> {status}",
              flush=True)
        for p in problems:
            print("      ", p, flush=True)
        failures += bool(problems)
    print(f"{rounds} rounds, {failures} with mismatches, {order_dependent} with order-dependent merges (valid, row count within 50 %), {time.time() - t0:.0f} s")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
