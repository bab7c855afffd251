"""Large single clusters for the parity tests of the whole-GPU EM route (test-only; vectorised numpy: a 200 000-row cluster
in well under a second).

Rows follow the invariants of ReadPathProbabilities (src/read_path_probabilities.cpp:184,212-219): every probability is
already multiplied by (1 - noise), at least prob_precision, ascending inside a row; every (probability, path) pair is a
group of its own.  The rows are NOT sorted and merged as the caller would (src/main.cpp:953-973): the estimators do not
rely on it, and a random cluster of this size has no identical rows.
"""
from __future__ import annotations

import numpy as np

from rpvg_amd.batch import ClusterBatch


def cluster_batch(rows: int, paths: int, per_row: int, seed: int, noise_only_frac: float = 0.0, max_count: int = 4,
                  groups: int = 1, haplotypes: int = 0) -> ClusterBatch:
    """One cluster of `rows` rows over `paths` paths, `per_row` distinct paths per row (all of them when per_row >= paths).
    A fraction of the rows carries no path at all (noise probability 1).  groups: transcripts (group_id) the paths are
    dealt over; haplotypes > 0: haplotype ids, each carrying one path of every transcript (source_ids)."""
    rng = np.random.default_rng(seed)
    per_row = min(per_row, paths)
    if per_row == paths:
        cols = np.broadcast_to(np.arange(paths, dtype=np.uint32), (rows, paths)).copy()
    else:
        # distinct paths per row: a random start and distinct strides over a permutation would correlate rows; draw and
        # repair duplicates instead (few when per_row << paths)
        cols = rng.integers(0, paths, size=(rows, per_row), dtype=np.int64)
        cols.sort(axis=1)
        for _ in range(64):
            dup = np.zeros_like(cols, dtype=bool)
            dup[:, 1:] = cols[:, 1:] == cols[:, :-1]
            if not dup.any():
                break
            cols[dup] = rng.integers(0, paths, size=int(dup.sum()))
            cols.sort(axis=1)
        else:
            raise RuntimeError("could not draw distinct paths")
        cols = cols.astype(np.uint32)
    # expression-like weights so that the EM has something to find
    theta = rng.lognormal(0.0, 1.5, size=paths)
    w = theta[cols] * (rng.random((rows, per_row)) + 0.05)
    noise = rng.choice(np.array([1e-4, 1e-3, 1e-2, 0.1]), size=rows, p=[0.7, 0.15, 0.1, 0.05])
    p = w / w.sum(axis=1, keepdims=True) * (1.0 - noise)[:, None]
    p = np.maximum(p, 1e-7)
    order = np.argsort(p, axis=1, kind="stable")
    p = np.take_along_axis(p, order, axis=1)
    cols = np.take_along_axis(cols, order, axis=1)
    count = rng.integers(1, max_count + 1, size=rows, dtype=np.uint32)
    has_paths = rng.random(rows) >= noise_only_frac
    noise = np.where(has_paths, noise, 1.0)
    n_ent = np.where(has_paths, per_row, 0).astype(np.uint64)
    row_grp_off = np.concatenate([[0], np.cumsum(n_ent)]).astype(np.uint64)
    keep = np.repeat(has_paths, per_row)
    grp_prob = p.reshape(-1)[keep]
    path_idx = cols.reshape(-1)[keep]
    G = int(row_grp_off[-1])
    group_id = (np.arange(paths) % max(1, groups)).astype(np.uint32)
    if haplotypes > 0:
        # every haplotype carries exactly one path of every transcript (tests/small_cases.py: the nested model relies on it)
        carried = [[] for _ in range(paths)]
        for g in range(max(1, groups)):
            members = np.nonzero(group_id == g)[0]
            for h in range(haplotypes):
                carried[int(rng.choice(members))].append(h)
        source_id = np.array([h for c in carried for h in c], dtype=np.uint32)
        source_off = np.concatenate([[0], np.cumsum([len(c) for c in carried])]).astype(np.uint64)
        source_count = np.array([max(1, len(c)) for c in carried], dtype=np.uint32)
    else:
        source_id = np.zeros(paths, dtype=np.uint32)
        source_off = np.arange(paths + 1, dtype=np.uint64)
        source_count = np.ones(paths, dtype=np.uint32)
    return ClusterBatch(
        cluster_row_off=np.array([0, rows], dtype=np.uint64), cluster_path_off=np.array([0, paths], dtype=np.uint64),
        row_count=count, row_noise=noise, row_grp_off=row_grp_off, grp_prob=grp_prob,
        grp_idx_off=np.arange(G + 1, dtype=np.uint64), path_idx=path_idx, path_group_id=group_id,
        path_source_count=source_count, path_source_off=source_off, source_id=source_id, path_effective_length=rng.uniform(200, 5000, size=paths))
