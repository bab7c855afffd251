"""Small seeded path clusters for parity tests (pure numpy/Python; test-only).

Miniature of the synthetic pantranscriptome of SURVEY.md §8d (S3): a cluster
has T transcripts (``group_id``), each with a few haplotype-specific
transcript (HST) paths; H haplotype ids, every haplotype carries exactly one
HST per transcript (``source_ids``); reads come from a true diplotype, are
compatible with the true HST plus a few sibling HSTs at a score deficit, and
are finished into rows the way ReadPathProbabilities::addPathProbs does
(src/read_path_probabilities.cpp:167-219), then sorted and merged the way the
caller does (src/main.cpp:953-973).
"""
from __future__ import annotations

import functools
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

SCORE_LOG_BASE = 1.383325268738  # src/utils.hpp:83
DOUBLE_PRECISION = np.finfo(np.float64).eps * 100


def _dc(a, b):
    return a == b or abs(a - b) < abs(min(a, b)) * DOUBLE_PRECISION


def finish_row(count: int, noise: float, lik: Dict[int, float], prec: float = 1e-8):
    """(count, noise, [(prob, [paths])...]) from per-path likelihoods."""
    tot = sum(lik.values())
    if not tot > 0:
        return (count, 1.0, [])
    buckets: List[List] = []
    low = 0.0
    for p in sorted(lik):
        pr = lik[p] / tot
        if pr < prec:
            low += pr
            continue
        for b in buckets:
            if abs(b[0] - pr) < prec:
                b[0] = (b[0] * len(b[1]) + pr) / (len(b[1]) + 1)
                b[1].append(p)
                break
        else:
            buckets.append([pr, [p]])
    groups = sorted((b[0] * (1 - noise), b[1]) for b in buckets)
    return (count, noise + low * (1 - noise), groups)


def _row_cmp(a, b):
    if not _dc(a[1], b[1]):
        return -1 if a[1] < b[1] else 1
    if len(a[2]) != len(b[2]):
        return -1 if len(a[2]) < len(b[2]) else 1
    for (pa, ia), (pb, ib) in zip(a[2], b[2]):
        if not _dc(pa, pb):
            return -1 if pa < pb else 1
        if len(ia) != len(ib):
            return -1 if len(ia) < len(ib) else 1
        if ia != ib:
            return -1 if ia < ib else 1
    if a[0] != b[0]:
        return -1 if a[0] < b[0] else 1
    return 0


def sort_and_merge(rows, prec: float = 1e-8):
    rows = sorted(rows, key=functools.cmp_to_key(_row_cmp))
    out = []
    for r in rows:
        if out:
            h = out[-1]
            same = abs(h[1] - r[1]) < prec and len(h[2]) == len(r[2]) and all(
                abs(x[0] - y[0]) < prec and x[1] == y[1] for x, y in zip(h[2], r[2]))
            if same:
                out[-1] = (h[0] + r[0], h[1], h[2])
                continue
        out.append(r)
    return out


def make_cluster(rng: np.random.Generator, n_transcripts: int = 2, hst_per_transcript: Sequence[int] = (3, 2),
                 n_haps: int = 8, n_reads: int = 300, tie_prob: float = 0.3, empty_read_frac: float = 0.02) -> dict:
    assert len(hst_per_transcript) == n_transcripts
    paths = []
    hst_of = []  # hst_of[t][h] = path index of haplotype h's HST of transcript t
    for t, n_hst in enumerate(hst_per_transcript):
        base = len(paths)
        assign = np.concatenate([np.arange(n_hst), rng.integers(0, n_hst, size=max(0, n_haps - n_hst))])[:n_haps]
        rng.shuffle(assign)
        if n_hst > n_haps:  # more HSTs than haplotypes: the tail carries no haplotype
            assign = rng.permutation(n_hst)[:n_haps]
        for j in range(n_hst):
            src = [int(h) for h in np.nonzero(assign == j)[0]]
            paths.append(dict(group_id=t, source_ids=src, source_count=max(1, len(src)),
                              effective_length=float(rng.uniform(200, 5000))))
        hst_of.append([base + int(assign[h]) for h in range(n_haps)])
    h1, h2 = (int(x) for x in rng.integers(0, n_haps, size=2))
    expr = rng.lognormal(0, 1, size=n_transcripts)
    expr /= expr.sum()
    allele_ratio = rng.beta(4, 4)
    mapq_noise = [1e-4, 1e-3, 0.1, 10 ** -0.3]
    rows = []
    for _ in range(n_reads):
        if rng.random() < empty_read_frac:
            rows.append((1, 1.0, []))
            continue
        t = int(rng.choice(n_transcripts, p=expr))
        true_path = hst_of[t][h1 if rng.random() < allele_ratio else h2]
        noise = mapq_noise[int(rng.choice(4, p=[0.70, 0.15, 0.10, 0.05]))]
        sibs = [p for p in range(len(paths)) if paths[p]["group_id"] == t and p != true_path]
        n_sib = min(len(sibs), int(rng.geometric(0.5)) - 1)
        lik = {true_path: 1.0 / paths[true_path]["effective_length"]}
        for p in (rng.choice(sibs, size=n_sib, replace=False) if n_sib else []):
            d = 0 if rng.random() < tie_prob else min(20, 1 + int(rng.poisson(3)))
            lik[int(p)] = math.exp(-SCORE_LOG_BASE * d) / paths[int(p)]["effective_length"]
        rows.append(finish_row(1, noise, lik))
    return dict(paths=paths, rows=sort_and_merge(rows))


def make_batch_clusters(seed: int, n_clusters: int = 6, max_reads: int = 400, with_empty: bool = True) -> List[dict]:
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_clusters):
        T = int(rng.integers(1, 4))
        hst = [int(rng.integers(1, 5)) for _ in range(T)]
        out.append(make_cluster(rng, T, hst, n_haps=int(rng.integers(2, 9)), n_reads=int(rng.integers(20, max_reads))))
    if with_empty:
        # a cluster without reads (src/path_abundance_estimator.cpp:20-22) and one whose reads all carry no path
        out.append(dict(paths=[dict(group_id=0, source_ids=[0], source_count=1, effective_length=100.0),
                               dict(group_id=0, source_ids=[1], source_count=1, effective_length=100.0)], rows=[]))
        out.append(dict(paths=[dict(group_id=0, source_ids=[0], source_count=1, effective_length=100.0),
                               dict(group_id=0, source_ids=[1], source_count=1, effective_length=100.0)],
                        rows=[(2, 1.0, []), (3, 1.0, [])]))
    return out


def rel_close(a, b, rel: float = 1e-4, floor: float = 1e-8) -> bool:
    """|a-b| <= rel*max(|a|,|b|) with an absolute floor (= prob_precision): the parity bar of BASELINE.json."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all(np.abs(a - b) <= np.maximum(rel * np.maximum(np.abs(a), np.abs(b)), floor)))


def build_reference_factory() -> str:
    """Compiles tests/cpp/reference_factory.cpp — the estimator factory of src/main.cpp:766-788 with the reference's
    constructor parameter lists — against the host library; returns the binary."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    binary = os.path.join(out_dir, "reference_factory")
    host, csrc = os.path.join(root, "rpvg_amd", "host"), os.path.join(root, "rpvg_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fopenmp", "-I" + host, os.path.join(root, "tests", "cpp", "reference_factory.cpp"),
                           "-o", binary, "-L" + host, "-lrpvg_amd_host", "-L" + csrc, "-lrpvg_hip", "-Wl,-rpath," + host, "-Wl,-rpath," + csrc])
    return binary
