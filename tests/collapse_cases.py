"""Clusters whose normalised group matrices hold rows within prob_precision of each other that the caller's merge did
not join — the inputs on which readCollapseProbabilityMatrix (src/path_estimator.cpp:197-259) changes results
(test-only helpers).

The caller merges adjacent rows of its sorted list (src/main.cpp:953-973); its order compares the noise probability
first (src/read_path_probabilities.cpp:283-322), so two rows that differ by 1e-9 in their noise end up far apart in
the list whenever rows with other path sets share the first one's noise.  They meet again in the group matrix, which
the estimators sort by column values (fuzz seed 7004, cluster 122: tests/golden/collapse_seed7004_cluster122.json).
"""
from __future__ import annotations

from typing import List

import numpy as np

from tests import small_cases


def plant_near_rows(rng: np.random.Generator, cluster: dict, fraction: float = 0.15, prec: float = 1e-8) -> dict:
    """Adds, for a share of the rows, one to three copies whose noise and / or one probability is moved by less than
    `prec`, then sorts and merges the way the caller does."""
    rows = list(cluster["rows"])
    extra = []
    for count, noise, groups in rows:
        if not groups or rng.random() >= fraction:
            continue
        for _ in range(int(rng.integers(1, 4))):
            new_noise = noise
            new_groups = [(p, list(ix)) for p, ix in groups]
            kind = int(rng.integers(0, 3))
            if kind != 1 and 1e-6 < noise < 0.9:
                new_noise = noise + float(rng.uniform(-0.9, 0.9)) * prec
            if kind != 0:
                g = int(rng.integers(0, len(new_groups)))
                p = new_groups[g][0] + float(rng.uniform(-0.9, 0.9)) * prec
                if p > 2 * prec:
                    new_groups[g] = (p, new_groups[g][1])
            new_groups.sort(key=lambda x: x[0])
            extra.append((int(rng.integers(1, 50)), new_noise, new_groups))
    return dict(paths=cluster["paths"], rows=small_cases.sort_and_merge(rows + extra, prec))


def make_collapse_clusters(seed: int, n_clusters: int = 8, max_reads: int = 300) -> List[dict]:
    rng = np.random.default_rng(seed)
    base = small_cases.make_batch_clusters(seed, n_clusters=n_clusters, max_reads=max_reads, with_empty=False)
    return [plant_near_rows(rng, c) for c in base]
