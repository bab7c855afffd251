"""Pins the CPU oracle (oracle/) before anything is compared against it.

* the reference's only test at this boundary, the weightedMinimumPathCover
  case of src/tests/path_abundance_estimator_test.cpp:8-28;
* the row-merge rule pinned by src/tests/read_path_probabilities_test.cpp:194-204;
* the hand-derivable known answers of SURVEY.md §8c (EM iteration counts and
  fixed points, permutation counts, single-read posteriors);
* the independent numpy restatement (oracle/np_oracle.py) on seeded clusters,
  for all three inference models;
* committed golden vectors (tests/golden/*.json).
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import np_oracle, pyoracle
from rpvg_amd.batch import ClusterBatch, make_params
from tests import small_cases

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


# ---- the reference's own golden vector --------------------------------------

def test_min_path_cover_reference_case():
    # src/tests/path_abundance_estimator_test.cpp:12-27
    cover = np.array([[1, 0, 1], [0, 1, 0], [1, 0, 0], [0, 1, 1]], dtype=np.uint8)
    counts = [1, 3, 1, 5]
    assert pyoracle.min_path_cover(cover, counts, [1, 1, 1]) == [0, 1]
    assert pyoracle.min_path_cover(cover, counts, [1, 1, 0.01]) == [0, 1, 2]


def test_min_path_cover_single_path():
    assert pyoracle.min_path_cover(np.array([[1], [1]], dtype=np.uint8), [1, 1], [1.0]) == [0]


# ---- scalar helpers -----------------------------------------------------------

def test_num_permutations():
    assert pyoracle.num_permutations([3]) == 1
    assert pyoracle.num_permutations([2, 2]) == 1
    assert pyoracle.num_permutations([1, 2]) == 2
    assert pyoracle.num_permutations([1, 1, 2]) == 3
    assert pyoracle.num_permutations([1, 2, 3]) == 6
    assert pyoracle.num_permutations([1, 1, 2, 2]) == 4  # n!/(n-u+1)!: kept as the reference has it
    for v in ([5], [1, 1], [0, 7], [2, 2, 9], [1, 1, 2, 2]):
        assert pyoracle.num_permutations(v) == np_oracle.num_permutations(v)


def test_add_log():
    lowest = -np.finfo(np.float64).max
    assert pyoracle.add_log(lowest, -3.0) == -3.0
    assert abs(pyoracle.add_log(math.log(0.25), math.log(0.5)) - math.log(0.75)) < 1e-15
    assert pyoracle.add_log(-1.5, -1.5) == np_oracle.add_log(-1.5, -1.5)


# ---- EM known answers -------------------------------------------

def test_kat_em_disjoint():
    n = 1e-4
    P = np.array([[1 - n, 0, n], [0, 1 - n, n]])
    ab, noise, total, its, _ = pyoracle.em_dense(P, [30, 70])
    assert its == 12
    assert total == 100
    assert small_cases.rel_close(ab, [30, 70], rel=1e-12)
    assert noise < 1e-30


def test_kat_em_tie():
    # the row pinned by src/tests/read_path_probabilities_test.cpp:29-34
    ab, noise, total, its, _ = pyoracle.em_dense(np.array([[0.45, 0.45, 0.1]]), [7])
    assert its == 21
    assert small_cases.rel_close(ab, [3.5, 3.5], rel=1e-12)
    assert noise < 1e-12
    assert abs(ab.sum() + noise - 7) < 1e-12


def test_kat_em_empty():
    ab, noise, total, its, _ = pyoracle.em_dense(np.array([[0, 0, 1.0], [0, 0, 1.0]]), [2, 3])
    assert its == 11
    assert list(ab) == [0, 0]
    assert noise == 5 and total == 5


def test_kat_em_single_path():
    ab, noise, total, its, _ = pyoracle.em_dense(np.array([[0.9, 0.1], [0.99, 0.01]]), [5, 5])
    assert its == 16
    assert abs(ab[0] - 10) < 1e-12 and noise < 1e-15


def test_em_float_start_value():
    # 1/float(C) start (src/path_abundance_estimator.cpp:54): with max_em_its=1 the result exposes it
    P = np.array([[0.5, 0.3, 0.2]])
    ab, noise, total, its, _ = pyoracle.em_dense(P, [1], max_em_its=1)
    a0 = np.float64(np.float32(1) / np.float32(3))
    expect = (P[0] * a0) / (P[0] * a0).sum()
    assert its == 1
    assert small_cases.rel_close(ab, expect[:2], rel=1e-15)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_em_matches_numpy_restatement(seed):
    rng = np.random.default_rng(seed)
    R, N = 60, 7
    P = rng.random((R, N)) * (rng.random((R, N)) < 0.4)
    noise = rng.choice([1e-4, 1e-3, 0.1], size=R)
    Pn = np_oracle.add_noise_and_normalize(P, noise)
    counts = rng.integers(1, 20, size=R).astype(np.float64)
    ab_c, nc_c, tot_c, its_c, _ = pyoracle.em_dense(Pn, counts)
    ab_n, nc_n, tot_n, its_n = np_oracle.em(Pn, counts)
    assert its_c == its_n
    assert tot_c == tot_n
    assert small_cases.rel_close(ab_c, ab_n, rel=1e-9)
    assert abs(nc_c - nc_n) <= 1e-9 * max(1.0, nc_n)
    assert abs(ab_c.sum() + nc_c - tot_c) < 1e-9 * tot_c  # PROP-mass


# ---- posteriors -----------------------------------------------------------------

def test_kat_posterior_single_read():
    # g = 1, one row: posterior_k ∝ (n + p_k)^c
    P = np.array([[0.6, 0.2]])
    sets, post = pyoracle.group_posteriors(P, [0.1], [3], [1, 1], 1)
    w = np.array([(0.1 + 0.6) ** 3, (0.1 + 0.2) ** 3])
    assert sets == [(0,), (1,)]
    assert small_cases.rel_close(post, w / w.sum(), rel=1e-13)


@pytest.mark.parametrize("seed", [11, 12, 13])
@pytest.mark.parametrize("g", [1, 2, 3])
def test_full_posteriors_match_numpy(seed, g):
    rng = np.random.default_rng(seed)
    R, N = 25, 5
    P = rng.random((R, N)) * (rng.random((R, N)) < 0.6) * 0.2
    noise = rng.choice([1e-4, 1e-2], size=R)
    counts = rng.integers(1, 6, size=R).astype(np.float64)
    pc = rng.integers(1, 5, size=N)
    sets_c, post_c = pyoracle.group_posteriors(P, noise, counts, pc, g)
    sets_n, post_n = np_oracle.posteriors_full(P, noise, counts, pc, g)
    assert sets_c == [tuple(s) for s in sets_n]
    assert small_cases.rel_close(post_c, post_n, rel=1e-10, floor=1e-300)
    assert abs(post_c.sum() - 1) < 1e-12


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_bounded_matches_numpy_and_full(seed):
    rng = np.random.default_rng(seed)
    R, N = 40, 6
    P = rng.random((R, N)) * (rng.random((R, N)) < 0.5) * 0.3
    noise = rng.choice([1e-4, 1e-3], size=R)
    counts = rng.integers(1, 8, size=R).astype(np.float64)
    pc = rng.integers(1, 4, size=N)
    sets_b, post_b = pyoracle.group_posteriors(P, noise, counts, pc, 2, bounded=True, min_rel_lik=1e-8)
    sets_n, post_n = np_oracle.posteriors_bounded(P, noise, counts, pc, 1e-8)
    assert sets_b == [tuple(s) for s in sets_n]
    assert small_cases.rel_close(post_b, post_n, rel=1e-10, floor=1e-300)
    # PROP-bounded: kept pairs agree with the exhaustive posterior up to the dropped mass
    sets_f, post_f = pyoracle.group_posteriors(P, noise, counts, pc, 2)
    full = {tuple(sorted(s)): p for s, p in zip(sets_f, post_f)}
    kept = {tuple(sorted(s)) for s, p in zip(sets_b, post_b) if p > 0}  # late-pruned pairs stay listed with posterior 0
    dropped = sum(p for s, p in full.items() if s not in kept)
    assert dropped < 1e-6
    for s, p in zip(sets_b, post_b):
        if p > 0:
            assert abs(p - full[tuple(sorted(s))]) <= 2 * dropped + 1e-12


# ---- estimators vs the numpy restatement ------------------------------------------

@pytest.mark.parametrize("seed", [101, 102])
def test_transcripts_model_matches_numpy(seed):
    clusters = small_cases.make_batch_clusters(seed)
    est, _ = pyoracle.run("transcripts", make_params(), ClusterBatch.from_clusters(clusters), 2)
    for cl, e in zip(clusters, est):
        ref = np_oracle.estimate_transcripts(cl["paths"], cl["rows"])
        assert e.path_group_sets == ref["sets"]
        assert e.em_iters == ref["iters"]
        assert e.total_count == ref["total"]
        assert small_cases.rel_close(e.abundances, ref["abund"], rel=1e-9)
        assert abs(e.noise_count - ref["noise"]) <= 1e-9 * max(1.0, ref["noise"])
        if cl["rows"]:
            assert abs(e.abundances.sum() + e.noise_count - e.total_count) <= 1e-9 * e.total_count


@pytest.mark.parametrize("seed", [201, 202, 203])
def test_haplotype_transcripts_model_matches_numpy(seed):
    clusters = small_cases.make_batch_clusters(seed)
    est, _ = pyoracle.run("haplotype-transcripts", make_params(), ClusterBatch.from_clusters(clusters), 2)
    for cl, e in zip(clusters, est):
        ref = np_oracle.estimate_haplotype_transcripts(cl["paths"], cl["rows"])
        got = e.keyed()
        assert set(got) == set(ref["keyed"])
        for key, (post, ab) in ref["keyed"].items():
            assert small_cases.rel_close(got[key][0], post, rel=1e-9)
            assert small_cases.rel_close(got[key][1], ab, rel=1e-8)
        assert e.total_count == ref["total"]
        assert abs(e.noise_count - ref["noise"]) <= 1e-8 * max(1.0, ref["total"])
        assert dict(zip(e.em_cols, e.em_iters)) == ref["iters"]
        if cl["rows"]:
            assert abs(e.abundances.sum() + e.noise_count - e.total_count) <= 1e-9 * e.total_count


@pytest.mark.parametrize("seed", [301, 302])
@pytest.mark.parametrize("ploidy", [1, 2, 3])
def test_haplotypes_model_matches_numpy(seed, ploidy):
    clusters = small_cases.make_batch_clusters(seed, n_clusters=4)
    est, _ = pyoracle.run("haplotypes", make_params(ploidy=ploidy), ClusterBatch.from_clusters(clusters), 2)
    for cl, e in zip(clusters, est):
        ref = np_oracle.estimate_haplotypes(cl["paths"], cl["rows"], ploidy)
        got = {k: v[0] for k, v in e.keyed().items()}
        assert set(got) == set(ref["keyed"])
        for key, post in ref["keyed"].items():
            assert small_cases.rel_close(got[key], post, rel=1e-9, floor=1e-300)
        # Full leaves zero-filled abundances behind (resetEstimates(n, g)), Bounded none
        assert not np.any(e.abundances)


def test_collapse_is_a_noop_up_to_precision():
    # PROP-collapse: rows duplicated with a < prob_precision perturbation merge; EM result and iterations unchanged
    rng = np.random.default_rng(5)
    R, N = 50, 4
    P = rng.random((R, N)) * (rng.random((R, N)) < 0.5)
    P[P.sum(axis=1) == 0, 0] = 0.5
    noise = rng.choice([1e-4, 1e-3], size=R)
    Pn = np_oracle.add_noise_and_normalize(P, noise)
    counts = rng.integers(1, 9, size=R).astype(np.float64)
    dup = np.concatenate([Pn, Pn * (1 + 3e-9 * rng.random((R, 1)))])
    dup_counts = np.concatenate([counts, counts])
    Pc, cc = np_oracle.read_collapse(dup, dup_counts, 1e-8)
    assert Pc.shape[0] <= R
    ab1, n1, t1, its1, _ = pyoracle.em_dense(dup, dup_counts)
    ab2, n2, t2, its2, _ = pyoracle.em_dense(Pc, cc)
    assert its1 == its2 and t1 == t2
    assert small_cases.rel_close(ab1, ab2, rel=1e-7)


# ---- committed golden vectors ---------------------------------------------------------

@pytest.mark.parametrize("name", ["transcripts", "haplotype-transcripts", "haplotypes"])
def test_oracle_reproduces_golden_vectors(name):
    path = os.path.join(GOLDEN, f"oracle_{name}.json")
    with open(path) as f:
        gold = json.load(f)
    clusters = [dict(paths=c["paths"], rows=[(r[0], r[1], [(g[0], g[1]) for g in r[2]]) for r in c["rows"]])
                for c in gold["clusters"]]
    est, _ = pyoracle.run(name, make_params(**gold["params"]), ClusterBatch.from_clusters(clusters), 1)
    for e, ge in zip(est, gold["estimates"]):
        got = e.keyed()
        want = {tuple(k): (p, tuple(a)) for k, p, a in ge["sets"]}
        assert set(got) == set(want)
        for key, (post, ab) in want.items():
            assert small_cases.rel_close(got[key][0], post, rel=1e-10, floor=1e-300)
            assert small_cases.rel_close(got[key][1], ab, rel=1e-10)
        assert e.total_count == ge["total_count"]
        assert abs(e.noise_count - ge["noise_count"]) <= 1e-10 * max(1.0, ge["total_count"])
        assert sorted(e.em_iters) == sorted(ge["em_iters"])


# ---- row collapse fixtures (tests/golden/make_collapse_fixture.py, tests/collapse_cases.py) ---------------------------

def test_collapse_fixture_is_what_the_oracle_computes():
    with open(os.path.join(GOLDEN, "collapse_seed7004_cluster122.json")) as f:
        doc = json.load(f)
    c = doc["cluster"]
    cluster = dict(paths=c["paths"], rows=[(r[0], r[1], [(g[0], g[1]) for g in r[2]]) for r in c["rows"]])
    batch = ClusterBatch.from_clusters([cluster])
    for case in doc["cases"]:
        est, _ = pyoracle.run(doc["model"], make_params(**case["params"]), batch, 1)
        keyed = est[0].keyed()
        want = {tuple(s[0]): (s[1], tuple(s[2])) for s in case["sets"]}
        assert set(keyed) == set(want)
        for key, (post, ab) in want.items():
            assert small_cases.rel_close(keyed[key][0], post, rel=1e-12) and small_cases.rel_close(keyed[key][1], ab, rel=1e-12)
    # the point of the fixture: without the collapse the haploid posterior of haplotype group 0 is 6.6715e-3, with it 6.6786e-3
    g, _ = np_oracle.source_groups(cluster["paths"])
    M, noise, counts = np_oracle.grouped_matrix(cluster["rows"], g)
    Pn = np_oracle.add_noise_and_normalize(M, noise)
    Pc, cc = np_oracle.read_collapse(Pn, counts, 1e-8)
    assert Pc.shape[0] < Pn.shape[0]
    haploid = {tuple(s[0]): s[1] for s in doc["cases"][0]["sets"]}
    assert abs(haploid[(0,)] - 6.6786e-3) < 1e-6


@pytest.mark.parametrize("seed", [811, 821])
def test_planted_close_rows_change_the_collapsed_matrix_and_both_oracles_agree(seed):
    from tests import collapse_cases
    clusters = collapse_cases.make_collapse_clusters(seed, n_clusters=6, max_reads=150)
    shrunk = 0
    for cl in clusters:
        g, _ = np_oracle.source_groups(cl["paths"])
        M, noise, counts = np_oracle.grouped_matrix(cl["rows"], g)
        Pn = np_oracle.add_noise_and_normalize(M, noise)
        Pc, cc = np_oracle.read_collapse(Pn, counts, 1e-8)
        assert cc.sum() == counts.sum()
        shrunk += Pn.shape[0] - Pc.shape[0]
    assert shrunk > 0
    batch = ClusterBatch.from_clusters(clusters)
    est, _ = pyoracle.run("haplotype-transcripts", make_params(), batch, 1)
    for cl, e in zip(clusters, est):
        ref = np_oracle.estimate_haplotype_transcripts(cl["paths"], cl["rows"])
        keyed = e.keyed()
        assert set(keyed) == set(ref["keyed"])
        for k, (p, a) in ref["keyed"].items():
            assert small_cases.rel_close(keyed[k][0], p, rel=1e-9) and small_cases.rel_close(keyed[k][1], a, rel=1e-8)
