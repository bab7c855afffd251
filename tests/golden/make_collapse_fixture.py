"""Generates tests/golden/collapse_seed7004_cluster122.json: the cluster on which the row collapse of the group
matrices (readCollapseProbabilityMatrix, src/path_estimator.cpp:197-259) first showed in the parity sweep
(tests/fuzz_parity.py seed 7004, cluster 122: a 40-read row with noise 1e-4 and a 1-read row with noise
1.0000266e-4 on the same path), together with what the CPU oracle makes of it under the sweep's options (ploidy 1,
three EM iterations) and under the reference's defaults.  Inputs and expected outputs only.

    python tests/golden/make_collapse_fixture.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import pyoracle  # noqa: E402
from rpvg_amd.batch import ClusterBatch, make_params  # noqa: E402
from tests import small_cases  # noqa: E402

SEED, CLUSTER = 7004, 122


def sweep_case(seed):
    """The draws of tests/fuzz_parity.py for a shape-0 seed."""
    rng = np.random.default_rng(seed)
    shape = rng.integers(0, 3)
    assert shape == 0
    clusters = small_cases.make_batch_clusters(seed, n_clusters=int(rng.integers(1, 200)), max_reads=int(rng.integers(25, 400)))
    model = ["transcripts", "haplotype-transcripts", "haplotypes", "strains"][int(rng.integers(0, 4))]
    kw = dict(max_em_its=int(rng.choice([3, 50, 10000])), max_rel_em_conv=float(rng.choice([1e-3, 1e-2, 1e-5])),
              min_hap_prob=float(rng.choice([1e-3, 1e-2, 1e-5])), rng_seed=int(rng.integers(0, 1000)))
    if model in ("haplotype-transcripts", "haplotypes"):
        kw["ploidy"] = int(rng.choice([1, 2, 2, 2, 3]))
        kw["use_hap_gibbs"] = int(rng.random() < 0.25)
    return clusters, model, kw


def main():
    clusters, model, kw = sweep_case(SEED)
    assert model == "haplotype-transcripts"
    cluster = clusters[CLUSTER]
    batch = ClusterBatch.from_clusters([cluster])
    cases = []
    for params in (kw, dict(), dict(ploidy=1)):
        est, _ = pyoracle.run(model, make_params(**params), batch, 1)
        e = est[0]
        cases.append(dict(params=params,
                          sets=[[list(k), v[0], list(v[1])] for k, v in sorted(e.keyed().items())],
                          noise_count=e.noise_count, total_count=e.total_count,
                          em=sorted([list(map(int, c)), int(i)] for c, i in zip(e.em_cols, e.em_iters))))
    doc = dict(origin=f"tests/fuzz_parity.py seed {SEED} cluster {CLUSTER}", model=model,
               cluster=dict(paths=cluster["paths"], rows=[[r[0], r[1], [[g[0], g[1]] for g in r[2]]] for r in cluster["rows"]]),
               cases=cases)
    path = os.path.join(HERE, f"collapse_seed{SEED}_cluster{CLUSTER}.json")
    with open(path, "w") as f:
        json.dump(doc, f)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
