"""Independent numpy restatement of the rpvg inference hot path.

TEST INFRASTRUCTURE, NOT PRODUCT CODE ("parity unpinned", see
oracle/rpvg_oracle.hpp).  Written from the algorithm descriptions
(SURVEY.md appendix D), deliberately in a different form from the C++
oracle (fused EM update, vectorised log-likelihoods, dictionary-based
nested bookkeeping) so that the two restatements check each other.
Only for small cases: dense matrices and pure-Python loops.

Reference being restated: src/path_abundance_estimator.cpp:18-114,344-750,
src/path_estimator.cpp:55-166,197-259,315-473, src/path_posterior_estimator.cpp:9-71.
"""
from __future__ import annotations

import functools
import itertools
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

LOWEST = -np.finfo(np.float64).max
DOUBLE_PRECISION = np.finfo(np.float64).eps * 100  # src/utils.hpp:81


def double_compare(a: float, b: float) -> bool:  # src/utils.hpp:87-93
    return a == b or abs(a - b) < abs(min(a, b)) * DOUBLE_PRECISION


def add_log(x: float, y: float) -> float:  # src/utils.hpp:300-302
    return x + math.log1p(math.exp(y - x)) if x > y else y + math.log1p(math.exp(x - y))


def num_permutations(values: Sequence[int]) -> int:  # src/utils.hpp:95-117
    n = len(values)
    if n == 1:
        return 1
    u = len(set(values))
    return int(math.gamma(n + 1) / math.gamma(n - u + 2))


# ---- matrices ---------------------------------------------------------------

def dense_matrix(rows, num_paths: int, cols: Sequence[int] = None):
    """src/path_estimator.cpp:55-113.  rows: [(count, noise, [(prob, [idx..])..])..]."""
    cols = list(range(num_paths)) if cols is None else list(cols)
    cmap = {p: j for j, p in enumerate(cols)}
    P = np.zeros((len(rows), len(cols)))
    for i, (_, _, groups) in enumerate(rows):
        for prob, idxs in groups:
            for p in idxs:
                if p in cmap:
                    P[i, cmap[p]] = prob
    noise = np.array([r[1] for r in rows], dtype=np.float64)
    counts = np.array([r[0] for r in rows], dtype=np.float64)
    return P, noise, counts


def grouped_matrix(rows, groups: Sequence[Sequence[int]]):
    """src/path_estimator.cpp:115-154."""
    P = np.zeros((len(rows), len(groups)))
    member = {}
    for g, paths in enumerate(groups):
        for p in paths:
            member.setdefault(p, []).append(g)
    for i, (_, _, rg) in enumerate(rows):
        for prob, idxs in rg:
            for p in idxs:
                for g in member.get(p, []):
                    P[i, g] += prob
    noise = np.array([r[1] for r in rows], dtype=np.float64)
    counts = np.array([r[0] for r in rows], dtype=np.float64)
    return P, noise, counts


def add_noise_and_normalize(P: np.ndarray, noise: np.ndarray) -> np.ndarray:
    """src/path_estimator.cpp:156-166."""
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.zeros(P.shape[0])
        for j in range(P.shape[1]):
            s = s + P[:, j]
        Q = (P / s[:, None]) * (1 - noise)[:, None]
    Q[np.isnan(Q)] = 0
    return np.concatenate([Q, noise[:, None]], axis=1)


def read_collapse(P: np.ndarray, counts: np.ndarray, prob_precision: float):
    """src/path_estimator.cpp:197-259 — tolerant sort, then merge runs against the run head."""

    def cmp(a, b):
        ra, rb = a[0], b[0]
        for x, y in zip(ra, rb):
            if not double_compare(x, y):
                return -1 if x < y else 1
        if not double_compare(a[1], b[1]):
            return -1 if a[1] < b[1] else 1
        return 0

    items = sorted([(tuple(P[i]), float(counts[i])) for i in range(P.shape[0])], key=functools.cmp_to_key(cmp))
    out_rows, out_counts = [list(items[0][0])], [items[0][1]]
    for row, c in items[1:]:
        if all(abs(h - x) < prob_precision for h, x in zip(out_rows[-1], row)):
            out_counts[-1] += c
        else:
            out_rows.append(list(row))
            out_counts.append(c)
    return np.array(out_rows), np.array(out_counts)


# ---- EM ---------------------------------------------------------------------

def em(P: np.ndarray, counts: np.ndarray, max_em_its: int = 10000, max_rel_em_conv: float = 1e-3):
    """Appendix D.1 (fused form).  Returns (abundances[C-1], noise_count, total, iterations)."""
    R, Cn = P.shape
    total = float(counts.sum())
    a = np.full(Cn, np.float64(np.float32(1) / np.float32(Cn)))
    prev = a.copy()
    conv, its = 0, 0
    for _ in range(max_em_its):
        its += 1
        s = P @ a
        a = a * ((counts / s) @ P) / total
        live = a >= 1e-8
        ok = bool(np.all(np.abs(a[live] - prev[live]) / a[live] <= max_rel_em_conv))
        if ok:
            conv += 1
            if conv == 10:
                break
        else:
            conv = 0
        prev = a.copy()
    ab = np.where(a[:-1] >= 1e-8, a[:-1] * total, 0.0)
    noise_count = float(a[-1] * total + (a[:-1][a[:-1] < 1e-8] * total).sum())
    return ab, noise_count, total, its


# ---- group posteriors ---------------------------------------------------------

def log_freqs(path_counts):
    pc = np.asarray(path_counts, dtype=np.float64)
    return np.log(pc / pc.sum())


def set_loglik(P, noise, counts, members, g):
    v = noise.copy()
    for m in members:
        v = v + P[:, m] / g
    return float(counts @ np.log(v))


def posteriors_full(P, noise, counts, path_counts, g: int):
    """Appendix D.2 / src/path_estimator.cpp:332-377."""
    N = P.shape[1]
    lf = log_freqs(path_counts)
    sets = list(itertools.combinations_with_replacement(range(N), g))
    ll = []
    lse = LOWEST
    for S in sets:
        v = set_loglik(P, noise, counts, S, g) + sum(lf[m] for m in S) + math.log(num_permutations(S))
        ll.append(v)
        lse = add_log(lse, v)
    return sets, np.exp(np.array(ll) - lse)


def posteriors_bounded(P, noise, counts, path_counts, min_rel_lik: float):
    """Appendix D.3 / src/path_estimator.cpp:379-473."""
    N = P.shape[1]
    lf = log_freqs(path_counts)
    thr = math.log(min_rel_lik)
    _, marg = posteriors_full(P, noise, counts, path_counts, 1)
    order = [i for _, i in sorted([(marg[i], i) for i in range(N)], reverse=True)]
    m = P.max(axis=1) / 2
    best = LOWEST
    kept, lls = [], []
    for ai, a in enumerate(order):
        base = noise + P[:, a] / 2
        bound = float(counts @ np.log(base + m)) + lf[a] + math.log(2)
        if bound - best < thr:
            continue
        for b in order[ai:]:
            v = float(counts @ np.log(base + P[:, b] / 2)) + lf[a] + lf[b] + math.log(num_permutations((a, b)))
            if v - best < thr:
                continue
            best = max(best, v)
            kept.append((a, b))
            lls.append(v)
    lse = LOWEST
    for i, v in enumerate(lls):
        if v - best < thr:
            lls[i] = LOWEST
        lse = add_log(lse, lls[i])
    with np.errstate(over="ignore", under="ignore"):
        post = np.exp(np.array(lls) - lse)
    return kept, post


# ---- estimators ---------------------------------------------------------------

def estimate_transcripts(paths, rows, max_em_its=10000, max_rel_em_conv=1e-3):
    """src/path_abundance_estimator.cpp:18-45 (-n 0)."""
    N = len(paths)
    if not rows:
        return dict(sets=[(i,) for i in range(N)], post=np.zeros(N), abund=np.zeros(N), noise=0.0, total=0.0, iters=[])
    P, noise, counts = dense_matrix(rows, N)
    Pn = add_noise_and_normalize(P, noise)
    ab, nc, total, its = em(Pn, counts, max_em_its, max_rel_em_conv)
    return dict(sets=[(i,) for i in range(N)], post=np.zeros(N), abund=ab, noise=nc, total=total, iters=[its])


def source_groups(paths):
    """src/path_abundance_estimator.cpp:493-546 — ascending source id order."""
    by_src: Dict[int, List[int]] = {}
    for i, p in enumerate(paths):
        for s in p["source_ids"]:
            by_src.setdefault(s, []).append(i)
    groups, mult = [], []
    seen: Dict[Tuple[int, ...], int] = {}
    for s in sorted(by_src):
        key = tuple(by_src[s])
        if key in seen:
            mult[seen[key]] += 1
        else:
            seen[key] = len(groups)
            groups.append(list(key))
            mult.append(1)
    return groups, mult


def estimate_haplotype_transcripts(paths, rows, ploidy=2, min_hap_prob=1e-3, prob_precision=1e-8,
                                   max_em_its=10000, max_rel_em_conv=1e-3):
    """Appendix D.5 / src/path_abundance_estimator.cpp:428-471,569-750 (collapsed, Bounded/Full, -n 0)."""
    if not rows:
        return dict(keyed={}, noise=0.0, total=0.0, iters={})
    groups, mult = source_groups(paths)
    G, noise, counts = grouped_matrix(rows, groups)
    Gn = add_noise_and_normalize(G, noise)
    Gn, gcounts = read_collapse(Gn, counts, prob_precision)
    gnoise, M = Gn[:, -1].copy(), Gn[:, :-1].copy()
    if ploidy == 2:
        sets, post = posteriors_bounded(M, gnoise, gcounts, mult, min_hap_prob)
    else:
        sets, post = posteriors_full(M, gnoise, gcounts, mult, ploidy)

    subsets: Dict[Tuple[int, ...], float] = {}
    tot_post = 0.0
    for S, pr in zip(sets, post):
        if pr >= min_hap_prob:
            key = tuple(sorted(p for g in S for p in groups[g]))
            subsets[key] = subsets.get(key, 0.0) + pr
            tot_post += pr
    for k in subsets:
        subsets[k] /= tot_post

    total = float(sum(r[0] for r in rows))
    keyed: Dict[Tuple[int, ...], list] = {}
    noise_count, sum_hap = 0.0, 0.0
    iters = {}
    for U in sorted(subsets):
        w = subsets[U]
        if w < min_hap_prob:
            continue
        sum_hap += w
        distinct = sorted(set(U))
        multiplicity = {p: U.count(p) for p in distinct}
        P, pn, pc = dense_matrix(rows, len(paths), distinct)
        Pn = add_noise_and_normalize(P, pn)
        Pn, pc = read_collapse(Pn, pc, prob_precision)
        ab, nc, _, its = em(Pn, pc, max_em_its, max_rel_em_conv)
        iters[tuple(distinct)] = its
        noise_count += nc * w
        by_group: Dict[int, List[int]] = {}
        for p in U:
            by_group.setdefault(paths[p]["group_id"], []).append(p)
        for gid in sorted(by_group):
            key = tuple(by_group[gid])
            ent = keyed.setdefault(key, [0.0, [0.0] * len(key)])
            ent[0] += w
            for i, p in enumerate(key):
                ent[1][i] += ab[distinct.index(p)] * w / multiplicity[p]
    noise_count += (1 - sum_hap) * total
    return dict(keyed={k: (v[0], tuple(v[1])) for k, v in keyed.items()}, noise=noise_count, total=total, iters=iters)


def estimate_haplotypes(paths, rows, ploidy=2):
    """src/path_posterior_estimator.cpp:35-71 (no Gibbs)."""
    if not rows:
        return dict(keyed={})
    P, noise, counts = dense_matrix(rows, len(paths))
    pc = [p["source_count"] for p in paths]
    if ploidy == 2:
        sets, post = posteriors_bounded(P, noise, counts, pc, 1e-8)
    else:
        sets, post = posteriors_full(P, noise, counts, pc, ploidy)
    return dict(keyed={tuple(s): float(p) for s, p in zip(sets, post)})
