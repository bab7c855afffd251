"""readCollapseProbabilityMatrix (src/path_estimator.cpp:197-259) on the GPU path: every row of a normalised group
matrix takes the values of the head of its run in the reference's tolerant row order (rpvg_amd/csrc/row_collapse.hip).

Checked at the matrix level against the numpy restatement (oracle/np_oracle.py) and end to end against the C++
oracle: the cluster of the parity sweep on which the collapse first showed (fuzz seed 7004, cluster 122 — posteriors
6.6715e-3 without the collapse against the reference's 6.6786e-3), clusters with planted near-equal rows, and a
trimmed fixed-seed run of the sweep itself.
"""
import json
import os

import numpy as np
import pytest

from oracle import np_oracle, pyoracle
from rpvg_amd import engine as eng_mod
from rpvg_amd.batch import ClusterBatch, make_params
from tests import collapse_cases, fuzz_parity, small_cases

pytestmark = pytest.mark.gpu

REL = 1e-6
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def engine():
    e = eng_mod.Engine(0)
    yield e
    e.close()


def _fixture():
    with open(os.path.join(GOLDEN, "collapse_seed7004_cluster122.json")) as f:
        doc = json.load(f)
    c = doc["cluster"]
    cluster = dict(paths=c["paths"], rows=[(r[0], r[1], [(g[0], g[1]) for g in r[2]]) for r in c["rows"]])
    return cluster, doc


def _collapsed_logliks(cluster, groups, prec):
    """Single-column and pair log-likelihood sums of the collapsed normalised matrix (numpy restatement)."""
    M, noise, counts = np_oracle.grouped_matrix(cluster["rows"], groups)
    Pn = np_oracle.add_noise_and_normalize(M, noise)
    plain = (Pn[:, :-1], Pn[:, -1], counts)
    Pc, cc = np_oracle.read_collapse(Pn, counts, prec)
    return plain, (Pc[:, :-1], Pc[:, -1], cc)


@pytest.mark.parametrize("seed", [811, 812, 813, 814])
def test_collapsed_group_matrices_match_numpy(hip_ctx, seed):
    clusters = collapse_cases.make_collapse_clusters(seed, n_clusters=10)
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    groups = [np_oracle.source_groups(cl["paths"])[0] for cl in clusters]
    dg = hip_ctx.groups(dev, list(range(len(clusters))), groups, True, collapse_precision=1e-8)
    replayed, replaced, whole, active = dg.collapse_info()
    assert replayed > 0 and replaced > 0 and whole == 0 and active >= replaced  # the planted rows reach the replay
    moved = 0
    for m, (cl, g) in enumerate(zip(clusters, groups)):
        (M0, n0, c0), (M1, n1, c1) = _collapsed_logliks(cl, g, 1e-8)
        G = len(g)
        pairs = [(a, b) for a in range(G) for b in range(a, G)]
        want = np.array([np_oracle.set_loglik(M1, n1, c1, p, 2) for p in pairs])
        plain = np.array([np_oracle.set_loglik(M0, n0, c0, p, 2) for p in pairs])
        got = dg.loglik([m] * len(pairs), pairs, 2.0)
        assert small_cases.rel_close(got, want, rel=1e-10, floor=1e-9), (m, np.max(np.abs(got - want)))
        moved += int(np.any(np.abs(plain - want) > 1e-7 * np.abs(want)))
        rm = M1.max(axis=1)  # row maxima follow the collapsed values (src/path_estimator.cpp:414)
        wantb = np.array([float(c1 @ np.log(n1 + M1[:, a] / 2 + rm / 2)) for a in range(G)])
        gotb = dg.loglik([m] * G, [[a, 0xFFFFFFFF] for a in range(G)], 2.0, add_rowmax=[1] * G)
        assert small_cases.rel_close(gotb, wantb, rel=1e-10, floor=1e-9)
    assert moved > 0  # the cases tell a collapsed matrix from an uncollapsed one
    dg.free()
    dev.free()


def test_matrices_without_close_rows_are_left_alone(hip_ctx):
    clusters = small_cases.make_batch_clusters(501, n_clusters=6, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    groups = [np_oracle.source_groups(cl["paths"])[0] for cl in clusters]
    plain = hip_ctx.groups(dev, list(range(len(clusters))), groups, True)
    coll = hip_ctx.groups(dev, list(range(len(clusters))), groups, True, collapse_precision=1e-8)
    for m, g in enumerate(groups):
        cols = [[a] for a in range(len(g))]
        assert np.array_equal(plain.loglik([m] * len(g), cols, 1.0), coll.loglik([m] * len(g), cols, 1.0))
    assert coll.collapse_info()[1] == 0
    plain.free()
    coll.free()
    dev.free()


def test_sweep_cluster_seed7004_matches_oracle(engine):
    cluster, doc = _fixture()
    batch = ClusterBatch.from_clusters([cluster])
    prep = engine.prepare(batch)
    for case in doc["cases"]:
        got, _ = engine.run(doc["model"], make_params(**case["params"]), prep)
        g = got[0]
        gk = g.keyed()
        want = {tuple(s[0]): (s[1], tuple(s[2])) for s in case["sets"]}
        assert set(gk) == set(want), case["params"]
        for key, (post, ab) in want.items():
            assert small_cases.rel_close(gk[key][0], post, rel=REL), (case["params"], key, gk[key][0], post)
            assert small_cases.rel_close(gk[key][1], ab, rel=REL), (case["params"], key, gk[key][1], ab)
        assert g.total_count == case["total_count"]
        assert abs(g.noise_count - case["noise_count"]) <= REL * max(1.0, case["total_count"])
        assert sorted([list(map(int, c)), int(i)] for c, i in zip(g.em_cols, g.em_iters)) == case["em"]  # exact EM iteration counts


@pytest.mark.parametrize("seed", [821, 822, 823])
@pytest.mark.parametrize("kw", [dict(), dict(ploidy=1), dict(ploidy=3), dict(use_hap_gibbs=1), dict(ind_hap_inference=1)],
                         ids=["diploid", "haploid", "triploid", "gibbs", "independent"])
def test_planted_close_rows_match_oracle(engine, seed, kw):
    clusters = collapse_cases.make_collapse_clusters(seed, n_clusters=12, max_reads=200)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params(**kw)
    ref, _ = pyoracle.run("haplotype-transcripts", params, batch, 2)
    got, _ = engine.run("haplotype-transcripts", params, engine.prepare(batch))
    # (only independent inference WITH Gibbs posteriors is statistical: fuzz_parity.run_case)
    assert not fuzz_parity.compare(got, ref)


def test_collapsed_rows_reach_the_sequential_search(engine):
    """RPVG_HIP_PAIR_TILES=0: the sequential kernels wait for the collapse on its stream like the tile kernel does."""
    cl = collapse_cases.make_collapse_clusters(823, n_clusters=12, max_reads=200)
    batch = ClusterBatch.from_clusters(cl)
    ref, _ = pyoracle.run("haplotype-transcripts", make_params(), batch, 2)
    os.environ["RPVG_HIP_PAIR_TILES"] = "0"
    try:
        got, _ = engine.run("haplotype-transcripts", make_params(), engine.prepare(batch))
    finally:
        os.environ.pop("RPVG_HIP_PAIR_TILES", None)
    assert not fuzz_parity.compare(got, ref)


# seeds of the sweep that once differed (5109: tied diplotype order; 7004: row collapse) + one of every shape / model
# ... 21123: the sum of 20 000 subset weights passes the reference's 100-ulp assertion on this side's rounding; 21204:
# the tie of 5109 with a zero-weight pair listed for whichever column is visited first
SWEEP_SEEDS = [5109, 7004, 21123, 21204, 1000, 1001, 1002, 1003, 1004, 1006, 1007, 1013, 1015, 1019, 1027, 1035, 1036, 1043, 1047]


@pytest.mark.parametrize("seed", SWEEP_SEEDS)
def test_trimmed_parity_sweep(engine, seed):
    case = fuzz_parity.draw_case(seed)
    problems = fuzz_parity.run_case(engine, case)
    assert not problems, (seed, case["model"], case["kw"], problems[:5])


def test_em_problem_rows_are_not_collapsed_and_what_that_costs(engine, monkeypatch):
    """The reference collapses the rows of every EM problem too (src/path_abundance_estimator.cpp:266,668); this path does
    not.  Sweep seed 20034 — one 26-path problem over 31 679 rows — is where it shows most in 1 700
    configurations: one abundance 1.2e-6 off (relative), everything else identical incl. the EM iteration counts.  The
    sweep's own bar (1e-6) fails on it; the project's (1e-4) holds with two orders to spare."""
    case = fuzz_parity.draw_case(20034)
    strict = fuzz_parity.run_case(engine, case)
    assert all("abundance" in p for p in strict) and len(strict) <= 2, strict[:5]
    monkeypatch.setattr(fuzz_parity, "REL", 1e-5)
    assert not fuzz_parity.run_case(engine, case)
