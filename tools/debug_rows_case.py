#!/usr/bin/env python3
"""Re-runs one case of tests/fuzz_rows.py (seed as argument) with full tracebacks."""
import sys
import traceback

import numpy as np

sys.path.insert(0, ".")
from oracle import pyoracle  # noqa: E402
from rpvg_amd import hip  # noqa: E402
from rpvg_amd.rows import AlignmentBatch, RowParams  # noqa: E402
from tests import test_hip_rows as T  # noqa: E402
from tests import test_row_construction as kat  # noqa: E402

seed = int(sys.argv[1])
ctx = hip.Context(0)
rng = np.random.default_rng(seed)
chains = bool(rng.random() < 0.3)
collapse = bool(rng.random() < 0.25)
wide = bool(rng.random() < 0.15)
single_end = bool(rng.random() < 0.3)
precision = float(rng.choice([1e-8, 1e-8, 1e-6, 1e-3, 0.05]))
min_noise = float(rng.choice([0.0, 1e-4, 1e-2]))
clusters = T.make_alignment_clusters(seed, n_clusters=int(rng.integers(2, 12)), max_paths=int(rng.integers(1, 120)),
                                     reads_per_cluster=int(rng.integers(1, 3000 if not wide else 80)), collapse=collapse, wide=wide, chains=chains)
batch = AlignmentBatch.from_clusters(clusters)
prm = RowParams(prob_precision=precision, min_noise_prob=min_noise, is_single_end=single_end,
                frag_length_log_prob=None if single_end else kat.frag_table())
print(dict(chains=chains, collapse=collapse, wide=wide, single_end=single_end, precision=precision, min_noise=min_noise))
ref, _ = pyoracle.build_rows(batch, prm, merge=False)
got, _, _ = ctx.build_rows(batch, prm, merge=False)
try:
    T.compare_unmerged(got, ref)
    print("unmerged: identical")
except AssertionError:
    traceback.print_exc()
    for name in ("cluster_row_off", "row_count", "row_grp_off", "grp_idx_off", "path_idx"):
        print(name, np.array_equal(getattr(got, name), getattr(ref, name)))
    bad = np.nonzero(~np.isclose(got.row_noise, ref.row_noise, rtol=T.REL, atol=0))[0]
    print("noise mismatches", len(bad), [(int(i), got.row_noise[i], ref.row_noise[i]) for i in bad[:5]])
    if len(got.grp_prob) == len(ref.grp_prob):
        badp = np.nonzero(~np.isclose(got.grp_prob, ref.grp_prob, rtol=T.REL, atol=0))[0]
        print("prob mismatches", len(badp), [(int(i), got.grp_prob[i], ref.grp_prob[i]) for i in badp[:5]])
ref_m, _ = pyoracle.build_rows(batch, prm, merge=True)
got_m, _, _ = ctx.build_rows(batch, prm, merge=True)
print("merged rows", got_m.num_rows, ref_m.num_rows)
try:
    if chains or precision > 1e-8:
        T.check_valid_merge(got_m, got, ref_m, count_tolerance=0.15)
    else:
        T.compare_merged(got_m, ref_m)
    print("merged: ok")
except AssertionError:
    traceback.print_exc()
