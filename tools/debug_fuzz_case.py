#!/usr/bin/env python3
"""Re-runs one shape-0 case of tests/fuzz_parity.py (seed as argument, optional repeat count) and prints what differs."""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import pyoracle  # noqa: E402
from rpvg_amd import engine as eng_mod  # noqa: E402
from rpvg_amd.batch import ClusterBatch, make_params  # noqa: E402
from tests import small_cases  # noqa: E402
from tests.fuzz_parity import compare  # noqa: E402

seed = int(sys.argv[1])
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
shape = rng.integers(0, 3)
assert shape == 0
batch = ClusterBatch.from_clusters(small_cases.make_batch_clusters(seed, n_clusters=int(rng.integers(1, 200)), max_reads=int(rng.integers(25, 400))))
model = ["transcripts", "haplotype-transcripts", "haplotypes", "strains"][int(rng.integers(0, 4))]
kw = dict(max_em_its=int(rng.choice([3, 50, 10000])), max_rel_em_conv=float(rng.choice([1e-3, 1e-2, 1e-5])),
          min_hap_prob=float(rng.choice([1e-3, 1e-2, 1e-5])), rng_seed=int(rng.integers(0, 1000)))
if model in ("haplotype-transcripts", "haplotypes"):
    kw["ploidy"] = int(rng.choice([1, 2, 2, 2, 3]))
    kw["use_hap_gibbs"] = int(rng.random() < 0.25)
print(model, kw, "clusters", batch.num_clusters)
params = make_params(**kw)
eng = eng_mod.Engine(0)
ref, _ = pyoracle.run(model, params, batch, 32)
for rep in range(repeats):
    got, _ = eng.run(model, params, eng.prepare(batch))
    problems = compare(got, ref)
    print("run", rep, "problems", len(problems)); [print("   ", p) for p in problems]; k = 122; print("gpu em", list(zip(got[k].em_cols, got[k].em_iters))); print("ref em", list(zip(ref[k].em_cols, ref[k].em_iters)))
