"""GPU parity tests of the C-ABI kernels (include/rpvg_hip.h) against the CPU oracle.

Bar (BASELINE.json north_star): integer results exact (read totals, EM
iteration counts), abundances / log-likelihoods within 1e-4 relative with an
absolute floor of prob_precision = 1e-8.  In practice FP64 end to end gives
~1e-12; the tests assert 1e-7 so that a real regression is loud.
"""
import json
import os

import numpy as np
import pytest

from oracle import np_oracle, pyoracle
from rpvg_amd.batch import ClusterBatch, make_params
from tests import small_cases

pytestmark = pytest.mark.gpu

REL = 1e-7  # far inside the 1e-4 budget
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _golden_clusters(name):
    with open(os.path.join(GOLDEN, f"oracle_{name}.json")) as f:
        gold = json.load(f)
    clusters = [dict(paths=c["paths"], rows=[(r[0], r[1], [(g[0], g[1]) for g in r[2]]) for r in c["rows"]])
                for c in gold["clusters"]]
    return clusters, gold


def _nonempty(clusters):
    return [k for k, c in enumerate(clusters) if c["rows"]]


# ---- sparse batched EM -----------------------------------------------------------

def test_em_solve_golden_transcripts(hip_ctx):
    clusters, gold = _golden_clusters("transcripts")
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    ks = _nonempty(clusters)
    cols = [list(range(len(clusters[k]["paths"]))) for k in ks]
    abund, noise, total, iters = hip_ctx.em_solve(dev, ks, cols)
    for i, k in enumerate(ks):
        ge = gold["estimates"][k]
        want = np.array([s[2][0] for s in sorted(ge["sets"], key=lambda s: s[0])])
        assert total[i] == ge["total_count"]
        assert int(iters[i]) == ge["em_iters"][0]
        assert small_cases.rel_close(abund[i], want, rel=REL)
        assert abs(noise[i] - ge["noise_count"]) <= REL * max(1.0, ge["total_count"])
        assert abs(abund[i].sum() + noise[i] - total[i]) <= 1e-9 * total[i]


@pytest.mark.parametrize("seed", [401, 402, 403])
def test_em_solve_matches_oracle_all_columns(hip_ctx, seed):
    clusters = small_cases.make_batch_clusters(seed, n_clusters=12)
    batch = ClusterBatch.from_clusters(clusters)
    est, _ = pyoracle.run("transcripts", make_params(), batch, 2)
    dev = hip_ctx.upload(batch)
    ks = _nonempty(clusters)
    cols = [list(range(len(clusters[k]["paths"]))) for k in ks]
    abund, noise, total, iters = hip_ctx.em_solve(dev, ks, cols)
    for i, k in enumerate(ks):
        e = est[k]
        assert total[i] == e.total_count
        assert [int(iters[i])] == e.em_iters
        assert small_cases.rel_close(abund[i], e.abundances, rel=REL)
        assert abs(noise[i] - e.noise_count) <= REL * max(1.0, e.total_count)


def test_register_em_in_copies_equals_one_copy_to_the_bit(hip_ctx):
    """emRegisterKernel<1,16> holds a problem of at most 16 / 32 rows four / two times and gives every copy a share of the
    columns in the M-step (the halving steps it leaves out would add zeros): abundances, noise and iteration counts equal
    RPVG_HIP_EM_COPIES=0 (every problem with one copy) to the bit; both equal the oracle."""
    rng = np.random.default_rng(451)
    clusters = []
    for reads in (1, 3, 9, 16, 17, 25, 32, 33, 50, 64):   # merged rows are at most the reads
        for paths in (1, 2, 7, 15):
            clusters.append(small_cases.make_cluster(rng, 1, [paths], n_haps=max(2, paths), n_reads=reads, empty_read_frac=0.1))
    batch = ClusterBatch.from_clusters(clusters)
    est, _ = pyoracle.run("transcripts", make_params(), batch, 2)
    dev = hip_ctx.upload(batch)
    ks = _nonempty(clusters)
    cols = [list(range(len(clusters[k]["paths"]))) for k in ks]
    results = []
    for copies in ("1", "0"):
        os.environ["RPVG_HIP_EM_COPIES"] = copies
        try:
            results.append(hip_ctx.em_solve(dev, ks, cols))
        finally:
            os.environ.pop("RPVG_HIP_EM_COPIES", None)
    (a1, n1, t1, i1), (a0, n0, t0, i0) = results
    assert np.array_equal(i1, i0) and np.array_equal(n1, n0) and np.array_equal(t1, t0)
    for x, y in zip(a1, a0):
        assert np.array_equal(x, y)
    for i, k in enumerate(ks):
        assert [int(i1[i])] == est[k].em_iters
        assert small_cases.rel_close(a1[i], est[k].abundances, rel=REL)


def test_em_solve_cluster_wider_than_lds(hip_ctx):
    """A cluster with more paths than LDS-resident abundance vectors hold (~9 700 columns): the reference's EM has no
    size limit (src/path_abundance_estimator.cpp:47-114); its vectors then live in global memory.  Next to small
    clusters in the same call."""
    rng = np.random.default_rng(431)
    n_paths, n_rows = 12000, 200
    paths = [dict(group_id=0, source_ids=[p], source_count=1, effective_length=1000.0) for p in range(n_paths)]
    rows = []
    for _ in range(n_rows):
        hit = sorted(int(x) for x in rng.choice(n_paths // 20, size=int(rng.integers(1, 4)), replace=False) * 20)
        lik = {p: float(rng.uniform(0.1, 1.0)) for p in hit}
        rows.append(small_cases.finish_row(int(rng.integers(1, 40)), 1e-4, lik))
    wide = dict(paths=paths, rows=small_cases.sort_and_merge(rows))
    clusters = small_cases.make_batch_clusters(432, n_clusters=3, with_empty=False) + [wide]
    batch = ClusterBatch.from_clusters(clusters)
    est, _ = pyoracle.run("transcripts", make_params(max_em_its=80), batch, 2)  # (the oracle's matrix is dense: 200 x 12 001)
    dev = hip_ctx.upload(batch)
    ks = list(range(len(clusters)))
    cols = [list(range(len(c["paths"]))) for c in clusters]
    abund, noise, total, iters = hip_ctx.em_solve(dev, ks, cols, max_em_its=80)
    for i, k in enumerate(ks):
        e = est[k]
        assert total[i] == e.total_count
        assert [int(iters[i])] == e.em_iters
        assert small_cases.rel_close(abund[i], e.abundances, rel=REL)
        assert abs(noise[i] - e.noise_count) <= REL * max(1.0, e.total_count)
    assert np.count_nonzero(abund[-1]) > 50 and int(iters[-1]) == 80


@pytest.mark.parametrize("seed", [411, 412])
def test_em_solve_column_subsets(hip_ctx, seed):
    """Subset problems (what haplotype-transcripts issues): oracle = Partial matrix -> normalise -> collapse -> EM."""
    rng = np.random.default_rng(seed)
    clusters = small_cases.make_batch_clusters(seed, n_clusters=8, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    ks, cols = [], []
    for k, cl in enumerate(clusters):
        n = len(cl["paths"])
        for _ in range(4):
            size = int(rng.integers(1, n + 1))
            ks.append(k)
            cols.append(sorted(int(x) for x in rng.choice(n, size=size, replace=False)))
    abund, noise, total, iters = hip_ctx.em_solve(dev, ks, cols)
    for i, (k, c) in enumerate(zip(ks, cols)):
        cl = clusters[k]
        P, pn, pc = np_oracle.dense_matrix(cl["rows"], len(cl["paths"]), c)
        Pn = np_oracle.add_noise_and_normalize(P, pn)
        Pn, pc = np_oracle.read_collapse(Pn, pc, 1e-8)
        ab, nc, tot, its, _ = pyoracle.em_dense(Pn, pc)
        assert total[i] == tot
        assert int(iters[i]) == its, (k, c)
        assert small_cases.rel_close(abund[i], ab, rel=REL)
        assert abs(noise[i] - nc) <= REL * max(1.0, tot)


def test_em_solve_edge_cases(hip_ctx):
    # SURVEY §8c known answers through the GPU path, in one ragged batch
    n = 1e-4
    clusters = [
        # KAT-EM-disjoint: 12 iterations, abundances (30, 70)
        dict(paths=[{}, {}], rows=[(30, n, [(1 - n, [0])]), (70, n, [(1 - n, [1])])]),
        # KAT-EM-tie: one row [0.45, 0.45 | 0.1] x 7 -> (3.5, 3.5), 21 iterations
        dict(paths=[{}, {}], rows=[(7, 0.1, [(0.45, [0, 1])])]),
        # KAT-EM-empty: only path-less rows -> (0, 0), noise 5, 11 iterations
        dict(paths=[{}, {}], rows=[(2, 1.0, []), (3, 1.0, [])]),
        # KAT-EM-single: one path
        dict(paths=[{}], rows=[(5, 0.1, [(0.9, [0])]), (5, 0.01, [(0.99, [0])])]),
    ]
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    abund, noise, total, iters = hip_ctx.em_solve(dev, [0, 1, 2, 3], [[0, 1], [0, 1], [0, 1], [0]])
    assert list(iters) == [12, 21, 11, 16]
    assert list(total) == [100, 7, 5, 10]
    assert small_cases.rel_close(abund[0], [30, 70], rel=1e-12)
    assert small_cases.rel_close(abund[1], [3.5, 3.5], rel=1e-12)
    assert list(abund[2]) == [0, 0] and noise[2] == 5
    assert abs(abund[3][0] - 10) < 1e-10
    # max_em_its is honoured exactly
    _, _, _, it1 = hip_ctx.em_solve(dev, [0, 1], [[0, 1], [0, 1]], max_em_its=3)
    assert list(it1) == [3, 3]


def test_em_solve_large_block_bins(hip_ctx):
    """A problem big enough for the 256- and 1024-thread kernels; oracle on the same rows."""
    rng = np.random.default_rng(77)
    n_paths = 300
    rows = []
    for _ in range(30000):
        k = int(rng.integers(1, 6))
        idx = sorted(int(x) for x in rng.choice(n_paths, size=k, replace=False))
        lik = {p: float(rng.random()) + 0.05 for p in idx}
        rows.append(small_cases.finish_row(int(rng.integers(1, 5)), float(rng.choice([1e-4, 1e-3, 0.1])), lik))
    cl = dict(paths=[{} for _ in range(n_paths)], rows=rows)
    small = dict(paths=[{} for _ in range(5)], rows=rows[:0] + [small_cases.finish_row(3, 1e-3, {0: 0.2, 3: 0.7})])
    batch = ClusterBatch.from_clusters([cl, small])
    est, _ = pyoracle.run("transcripts", make_params(max_em_its=200), batch, 2)
    dev = hip_ctx.upload(batch)
    abund, noise, total, iters = hip_ctx.em_solve(dev, [0, 1], [list(range(n_paths)), list(range(5))], max_em_its=200)
    for i in range(2):
        assert total[i] == est[i].total_count
        assert [int(iters[i])] == est[i].em_iters
        assert small_cases.rel_close(abund[i], est[i].abundances, rel=REL)


# ---- dense streaming EM -------------------------------------------------------------

def test_dense_from_cluster_and_em_dense(hip_ctx):
    clusters, gold = _golden_clusters("transcripts")
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    for k in _nonempty(clusters):
        cl = clusters[k]
        R, N = len(cl["rows"]), len(cl["paths"])
        ld = (N + 2) & ~1
        d_P, d_c = hip_ctx.malloc(R * ld * 8), hip_ctx.malloc(R * 8)
        try:
            total = hip_ctx.dense_from_cluster(dev, k, d_P, ld, d_c)
            P = hip_ctx.d2h(d_P, (R, ld))
            Pd, pn, pc = np_oracle.dense_matrix(cl["rows"], N)
            Pn = np_oracle.add_noise_and_normalize(Pd, pn)
            assert np.allclose(P[:, :N + 1], Pn, rtol=1e-14, atol=0)  # same two roundings per entry; row sums may differ by an ulp
            assert np.array_equal(P[:, :N + 1] == 0, Pn == 0)
            assert total == pc.sum()
            ab, noise, its = hip_ctx.em_dense(d_P, R, N + 1, ld, d_c, total)
            ge = gold["estimates"][k]
            want = np.array([s[2][0] for s in sorted(ge["sets"], key=lambda s: s[0])])
            assert its == ge["em_iters"][0]
            assert small_cases.rel_close(ab, want, rel=REL)
            assert abs(noise - ge["noise_count"]) <= REL * max(1.0, total)
        finally:
            hip_ctx.free(d_P)
            hip_ctx.free(d_c)


@pytest.mark.parametrize("R,N", [(3000, 2000), (5000, 130), (777, 1), (4096, 257)])
def test_synth_dense_em_matches_oracle(hip_ctx, R, N):
    ld = (N + 2) & ~1
    d_P, d_c = hip_ctx.malloc(R * ld * 8), hip_ctx.malloc(R * 8)
    try:
        hip_ctx.synth_dense_cluster(2, R, N, d_P, ld, d_c)
        P = hip_ctx.d2h(d_P, (R, ld))[:, :N + 1].copy()
        counts = hip_ctx.d2h(d_c, (R,))
        assert np.all(counts == 1)
        assert np.all(np.abs(P.sum(axis=1) - 1) < 1e-12)  # K2 invariant: paths + noise = 1
        assert np.all(P[:, :N] > 0)  # dense: every entry non-zero
        max_its = 40
        ab, noise, its = hip_ctx.em_dense(d_P, R, N + 1, ld, d_c, float(R), max_em_its=max_its)
        ab_o, noise_o, tot_o, its_o, _ = pyoracle.em_dense(P, counts, max_em_its=max_its)
        assert its == its_o
        assert small_cases.rel_close(ab, ab_o, rel=REL)
        assert abs(noise - noise_o) <= REL * R
        assert abs(ab.sum() + noise - R) <= 1e-9 * R
    finally:
        hip_ctx.free(d_P)
        hip_ctx.free(d_c)


def test_em_dense_converges_like_oracle(hip_ctx):
    R, N = 2000, 40
    ld = (N + 2) & ~1
    d_P, d_c = hip_ctx.malloc(R * ld * 8), hip_ctx.malloc(R * 8)
    try:
        hip_ctx.synth_dense_cluster(9, R, N, d_P, ld, d_c)
        P = hip_ctx.d2h(d_P, (R, ld))[:, :N + 1].copy()
        ab, noise, its = hip_ctx.em_dense(d_P, R, N + 1, ld, d_c, float(R))
        ab_o, noise_o, _, its_o, _ = pyoracle.em_dense(P, np.ones(R))
        assert its == its_o and its < 10000
        assert small_cases.rel_close(ab, ab_o, rel=REL)
    finally:
        hip_ctx.free(d_P)
        hip_ctx.free(d_c)


def test_synth_dense_rows_are_slices_of_the_cluster(hip_ctx):
    R, N = 1001, 70
    ld = (N + 2) & ~1
    d_P, d_c = hip_ctx.malloc(R * ld * 8), hip_ctx.malloc(R * 8)
    try:
        hip_ctx.synth_dense_cluster(4, R, N, d_P, ld, d_c)
        whole = hip_ctx.d2h(d_P, (R, ld)).copy()
        for r0, r1 in [(0, 500), (500, 1001), (333, 334)]:
            hip_ctx.synth_dense_rows(4, r0, r1 - r0, N, d_P, ld, d_c)
            part = hip_ctx.d2h(d_P, (r1 - r0, ld))
            assert np.array_equal(part, whole[r0:r1])
    finally:
        hip_ctx.free(d_P)
        hip_ctx.free(d_c)


@pytest.mark.parametrize("R,N", [(6000, 1500), (4000, 90)])
def test_one_rank_communicator_row_sharded_em(hip_ctx, R, N):
    """RCCL plumbing of the row-sharded dense EM with a one-rank communicator: unique id, init, the in-stream
    all-reduce of every iteration; the result must equal the unsharded solve bit for bit.  (The two-rank algebra
    is covered on CPU in tests/test_distributed_cpu.py; N > 1 GPUs are the round-end driver's to launch.)"""
    from rpvg_amd import hip
    ld = (N + 2) & ~1
    ctx = hip.Context(0)
    d_P = d_c = d_x = None
    try:
        d_P, d_c = ctx.malloc(R * ld * 8), ctx.malloc(R * 8)
        ctx.synth_dense_cluster(6, R, N, d_P, ld, d_c)
        with pytest.raises(hip.EngineError, match="communicator"):
            ctx.em_dense(d_P, R, N + 1, ld, d_c, float(R), max_em_its=30, sharded=True)
        ctx.comm_init(hip.Context.comm_unique_id(), 1, 0)
        with pytest.raises(hip.EngineError, match="already"):
            ctx.comm_init(hip.Context.comm_unique_id(), 1, 0)
        x = np.arange(1000, dtype=np.float64) * 0.25
        d_x = ctx.malloc(x.nbytes)
        ctx.h2d(d_x, x)
        ctx.allreduce_sum_f64(d_x, x.size)
        assert np.array_equal(ctx.d2h(d_x, (x.size,)), x)
        ab_s, noise_s, its_s = ctx.em_dense(d_P, R, N + 1, ld, d_c, float(R), max_em_its=30, sharded=True)
        ab, noise, its = ctx.em_dense(d_P, R, N + 1, ld, d_c, float(R), max_em_its=30)
        assert its_s == its and noise_s == noise and np.array_equal(ab_s, ab)
        P = ctx.d2h(d_P, (R, ld))[:, :N + 1].copy()
        ab_o, noise_o, _, its_o, _ = pyoracle.em_dense(P, np.ones(R), max_em_its=30)
        assert its_s == its_o and small_cases.rel_close(ab_s, ab_o, rel=REL)
        # the final gather (rpvg_hip_gather, ncclAllGather) in a world of one, and its argument checks
        assert np.array_equal(ctx.gather(x, [x.size]), x)
        with pytest.raises(hip.EngineError, match="counts"):
            ctx.gather(x, [x.size + 1])
        ctx.comm_destroy()
        ctx.comm_destroy()  # idempotent
        assert np.array_equal(ctx.gather(x, [x.size]), x)  # no communicator: a world of one
        ctx.comm_init(hip.Context.comm_unique_id(), 1, 0)
        # a rank without rows (fewer rows than ranks) still joins every all-reduce with zero column sums
        ab_e, noise_e, its_e = ctx.em_dense(d_P, 0, N + 1, ld, d_c, float(R), max_em_its=3, sharded=True)
        assert its_e == 3 and np.all(ab_e == 0)
        # the status words of the collective protocol: a rank that fails its local checks says so in a handshake before
        # the EM, a rank whose column sums are not finite raises the word that travels with them — in both cases the call
        # returns an error on every rank instead of leaving the peers in the next all-reduce
        os.environ["RPVG_HIP_INJECT_SHARD_FAILURE"] = "checks"
        try:
            with pytest.raises(hip.EngineError, match="injected failure of the local checks"):
                ctx.em_dense(d_P, R, N + 1, ld, d_c, float(R), max_em_its=30, sharded=True)
            os.environ["RPVG_HIP_INJECT_SHARD_FAILURE"] = "nan"
            with pytest.raises(hip.EngineError, match="not finite in EM iteration 1"):
                ctx.em_dense(d_P, R, N + 1, ld, d_c, float(R), max_em_its=30, sharded=True)
        finally:
            os.environ.pop("RPVG_HIP_INJECT_SHARD_FAILURE", None)
        ab_b, noise_b, its_b = ctx.em_dense(d_P, R, N + 1, ld, d_c, float(R), max_em_its=30, sharded=True)
        assert its_b == its and np.array_equal(ab_b, ab)
        ctx.comm_destroy()
        ctx.comm_init_all()  # the single-process form (one host thread per GPU): this context is rank 0 of 1
        ab_a, noise_a, its_a = ctx.em_dense(d_P, R, N + 1, ld, d_c, float(R), max_em_its=30, sharded=True)
        assert its_a == its and noise_a == noise and np.array_equal(ab_a, ab)
        ctx.comm_destroy()
    finally:
        for d in (d_P, d_c, d_x):
            if d:
                ctx.free(d)
        ctx.close()


@pytest.mark.parametrize("use_table", [False, True])
def test_device_logarithm_accuracy(hip_ctx, use_table):
    """The FP64 log of the log-likelihood kernels against extended precision: at most ~1 ulp over the arguments the
    kernels see (probabilities in (1e-8, 1], and up to 2 for the un-normalised raw matrices), and an absolute error
    below 2e-16 right above 1 where the table version trades relative for absolute accuracy."""
    rng = np.random.default_rng(77)
    x = np.concatenate([10.0 ** rng.uniform(-9, 0, 200000), rng.uniform(0.4, 1.0, 200000), rng.uniform(1.0, 2.0, 100000),
                        1 - 10.0 ** rng.uniform(-16, -3, 20000), 1 + 10.0 ** rng.uniform(-16, -3, 20000),
                        [1.0, 0.5, 0.25, 2.0, 1e-8, 0.70710678118654752]])
    got = hip_ctx.debug_log(x, use_table)
    ref_ld = np.log(x.astype(np.longdouble))
    ref = ref_ld.astype(np.float64)
    err = np.abs(got.astype(np.longdouble) - ref_ld).astype(np.float64)
    ulp = np.spacing(np.maximum(np.abs(ref), 1e-300))
    away = (x < 1.0) | (x > 1.002)
    assert float((err[away] / ulp[away]).max()) < 1.6, float((err[away] / ulp[away]).max())
    assert float(err.max()) < 2.2e-16 * max(1.0, float(np.abs(ref).max()))
    assert abs(got[np.where(x == 1.0)[0][0]]) < 1e-19


# ---- group log-likelihoods ------------------------------------------------------------

@pytest.mark.parametrize("normalise", [False, True])
def test_group_loglik_matches_numpy(hip_ctx, normalise):
    clusters = small_cases.make_batch_clusters(501, n_clusters=6, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    mats, groups = [], []
    for k, cl in enumerate(clusters):
        if normalise:
            g, _ = np_oracle.source_groups(cl["paths"])
        else:
            g = [[p] for p in range(len(cl["paths"]))]
        mats.append(k)
        groups.append(g)
    dg = hip_ctx.groups(dev, mats, groups, normalise)
    rng = np.random.default_rng(3)
    for m, (k, g) in enumerate(zip(mats, groups)):
        cl = clusters[k]
        M, noise, counts = np_oracle.grouped_matrix(cl["rows"], g)
        if normalise:
            Mn = np_oracle.add_noise_and_normalize(M, noise)
            M = Mn[:, :-1]
        G = len(g)
        pairs = [(a, b) for a in range(G) for b in range(a, G)]
        want2 = np.array([np_oracle.set_loglik(M, noise, counts, p, 2) for p in pairs])
        got2 = dg.loglik([m] * len(pairs), pairs, 2.0)
        assert small_cases.rel_close(got2, want2, rel=1e-11, floor=1e-9)
        want1 = np.array([np_oracle.set_loglik(M, noise, counts, (a,), 1) for a in range(G)])
        got1 = dg.loglik([m] * G, [[a] for a in range(G)], 1.0)
        assert small_cases.rel_close(got1, want1, rel=1e-11, floor=1e-9)
        # optimistic bound of the diploid search: noise + M_a/2 + rowmax/2
        rm = M.max(axis=1)
        wantb = np.array([float(counts @ np.log(noise + M[:, a] / 2 + rm / 2)) for a in range(G)])
        gotb = dg.loglik([m] * G, [[a, 0xFFFFFFFF] for a in range(G)], 2.0, add_rowmax=[1] * G)
        assert small_cases.rel_close(gotb, wantb, rel=1e-11, floor=1e-9)


@pytest.mark.parametrize("width", [1, 2, 3])
def test_group_conditionals_equal_member_list_requests(hip_ctx, width):
    """rpvg_hip_group_conditionals (one request per Gibbs conditional) returns, for every candidate column, exactly
    what rpvg_hip_group_loglik returns for the member list (others..., candidate)."""
    clusters = small_cases.make_batch_clusters(503, n_clusters=7, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    mats = list(range(len(clusters)))
    groups = [[[p] for p in range(len(cl["paths"]))] for cl in clusters]
    num_cols = [len(g) for g in groups]
    dg = hip_ctx.groups(dev, mats, groups, False)
    rng = np.random.default_rng(17)
    req_m = [int(m) for m in rng.integers(0, len(mats), size=23)]
    req_o = [[int(x) for x in rng.integers(0, num_cols[m], size=width - 1)] for m in req_m]
    got = dg.conditionals(req_m, req_o, width, float(width), num_cols)
    assert len(got) == len(req_m)
    for m, o, g in zip(req_m, req_o, got):
        G = num_cols[m]
        assert g.shape == (G,)
        want = dg.loglik([m] * G, [o + [k] for k in range(G)], float(width))
        assert np.allclose(g, want, rtol=1e-13, atol=0)  # same sums, different association (product chains)
    cl = clusters[req_m[0]]
    M, noise, counts = np_oracle.grouped_matrix(cl["rows"], groups[req_m[0]])
    want0 = np.array([np_oracle.set_loglik(M, noise, counts, tuple(req_o[0]) + (k,), width) for k in range(num_cols[req_m[0]])])
    assert small_cases.rel_close(got[0], want0, rel=1e-11, floor=1e-9)


@pytest.mark.parametrize("normalise", [False, True])
def test_single_path_matrices_equal_the_matrices_of_listed_single_path_columns(hip_ctx, normalise):
    """rpvg_hip_groups_build_single_paths: the matrices of the raw path posteriors (src/path_posterior_estimator.cpp:9-31, 35-71: a
    group per path) without their column lists crossing the ABI — the same log-likelihoods and conditionals, bit for bit, as from
    rpvg_hip_groups_build with one single-entry list per path; clusters of a few and of thousands of paths, listed twice and out of order."""
    from rpvg_amd import synth
    small = ClusterBatch.from_clusters(small_cases.make_batch_clusters(505, n_clusters=9, with_empty=False))
    wide = synth.generate(seed=12, num_clusters=3, total_paths=3000, total_reads=6000, max_cluster_paths=2500)
    batch = ClusterBatch.concat([small, wide])
    dev = hip_ctx.upload(batch)
    mats = [k for k in range(batch.num_clusters) if batch.cluster_row_off[k + 1] > batch.cluster_row_off[k]]
    mats = mats[::-1] + mats[:2]
    num_cols = [int(batch.cluster_path_off[k + 1] - batch.cluster_path_off[k]) for k in mats]
    listed = hip_ctx.groups(dev, mats, [[[p] for p in range(n)] for n in num_cols], normalise)
    implied = hip_ctx.groups(dev, mats, None, normalise)
    rng = np.random.default_rng(5)
    req_m = [int(m) for m in rng.integers(0, len(mats), size=40)]
    pairs = [[int(rng.integers(0, num_cols[m])), int(rng.integers(0, num_cols[m]))] for m in req_m]
    assert np.array_equal(listed.loglik(req_m, pairs, 2.0), implied.loglik(req_m, pairs, 2.0))
    singles = [[p[0]] for p in pairs]
    assert np.array_equal(listed.loglik(req_m, singles, 1.0), implied.loglik(req_m, singles, 1.0))
    others = [[p[0]] for p in pairs]
    for a, b in zip(listed.conditionals(req_m, others, 2, 2.0, num_cols), implied.conditionals(req_m, others, 2, 2.0, num_cols)):
        assert np.array_equal(a, b)
    dev.free()


def test_stats_report_kernel_time(hip_ctx):
    clusters = small_cases.make_batch_clusters(9, n_clusters=4, with_empty=False)
    dev = hip_ctx.upload(ClusterBatch.from_clusters(clusters))
    hip_ctx.reset_stats()
    hip_ctx.em_solve(dev, [0, 1], [list(range(len(clusters[0]["paths"]))), list(range(len(clusters[1]["paths"])))])
    st = hip_ctx.stats()
    assert st["em_sparse_launches"] >= 1 and st["em_sparse_ms"] > 0 and st["em_sparse_alg_bytes"] > 0
    assert st["em_iterations_total"] > 0


# ---- on-device diploid branch-and-bound ---------------------------------------------------

@pytest.mark.parametrize("tiles", [0, 2], ids=["sequential", "all-pairs"])
@pytest.mark.parametrize("normalise,thr", [(True, 1e-3), (False, 1e-8)])
def test_bounded_search_on_device_matches_oracle(hip_ctx, normalise, thr, tiles):
    """all-pairs (the default): every pair from LDS-staged rows (pairTile2Kernel: ranges of tiles, LDS-direct loads; 1: round 2's
    pairTileKernel) + prefix-maximum filter instead of the sequential in-workgroup walk (RPVG_HIP_PAIR_TILES=0) — the kept pairs
    and their order are the reference's either way."""
    rng = np.random.default_rng(701)
    clusters = small_cases.make_batch_clusters(702, n_clusters=10, with_empty=False)
    clusters.append(small_cases.make_cluster(rng, 3, [9, 7, 8], n_haps=40, n_reads=1500))
    clusters.append(small_cases.make_cluster(rng, 1, [30], n_haps=60, n_reads=3000))
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    mats, groups, counts = [], [], []
    for k, cl in enumerate(clusters):
        if normalise:
            g, mult = np_oracle.source_groups(cl["paths"])
        else:
            g = [[p] for p in range(len(cl["paths"]))]
            mult = [p["source_count"] for p in cl["paths"]]
        mats.append(k)
        groups.append(g)
        counts.append(mult)
    dg = hip_ctx.groups(dev, mats, groups, normalise)
    if tiles != 2:
        os.environ["RPVG_HIP_PAIR_TILES"] = str(tiles)
    if tiles == 0:
        os.environ["RPVG_HIP_PAIRS_WITH_COUNTS"] = "0"  # ... and the kept pairs fetched with a copy of their own
    try:
        got = dg.bounded_pair_posteriors(np.concatenate(counts), thr)
    finally:
        os.environ.pop("RPVG_HIP_PAIR_TILES", None)
        os.environ.pop("RPVG_HIP_PAIRS_WITH_COUNTS", None)
    for m, (k, g) in enumerate(zip(mats, groups)):
        cl = clusters[k]
        M, noise, cnts = np_oracle.grouped_matrix(cl["rows"], g)
        if normalise:
            M = np_oracle.add_noise_and_normalize(M, noise)[:, :-1]
        sets, post = pyoracle.group_posteriors(M, noise, cnts, counts[m], 2, bounded=True, min_rel_lik=thr)
        # same sequential search -> same kept pairs in the same order
        assert got[m][0] == sets, (m, len(got[m][0]), len(sets))
        assert small_cases.rel_close(got[m][1], post, rel=1e-9, floor=1e-300)
        assert abs(got[m][1].sum() - 1) < 1e-9


@pytest.mark.parametrize("force_table", [True, False, "tiles"])
def test_bounded_search_table_path(hip_ctx, force_table):
    """Big matrices are resolved from a parallel pair table + prefix-max filter instead of the
    in-workgroup sequential walk; both must give the reference's kept pairs, in its order.  "tiles": the default search
    (every pair of every matrix by pairTileKernel); the other two run the sequential kernels (RPVG_HIP_PAIR_TILES=0)."""
    rng = np.random.default_rng(711)
    clusters = [small_cases.make_cluster(rng, 3, [9, 7, 8], n_haps=40, n_reads=12000),
                small_cases.make_cluster(rng, 2, [5, 4], n_haps=16, n_reads=600)]
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    groups, counts = [], []
    for cl in clusters:
        g, mult = np_oracle.source_groups(cl["paths"])
        groups.append(g)
        counts.append(mult)
    dg = hip_ctx.groups(dev, [0, 1], groups, True)
    if force_table is True:
        os.environ["RPVG_HIP_TABLE_MIN_WORK"] = "0"
    if force_table != "tiles":
        os.environ["RPVG_HIP_PAIR_TILES"] = "0"
    try:
        got = dg.bounded_pair_posteriors(np.concatenate(counts), 1e-3)
    finally:
        os.environ.pop("RPVG_HIP_TABLE_MIN_WORK", None)
        os.environ.pop("RPVG_HIP_PAIR_TILES", None)
    for m, cl in enumerate(clusters):
        M, noise, cnts = np_oracle.grouped_matrix(cl["rows"], groups[m])
        M = np_oracle.add_noise_and_normalize(M, noise)[:, :-1]
        sets, post = pyoracle.group_posteriors(M, noise, cnts, counts[m], 2, bounded=True, min_rel_lik=1e-3)
        assert got[m][0] == sets
        assert small_cases.rel_close(got[m][1], post, rel=1e-9, floor=1e-300)


@pytest.mark.parametrize("haplotypes,reads,chunk_rows", [(64, 2600, None), (52, 1500, None), (100, 2300, None), (150, 1200, None), (64, 2600, 512)])
def test_tile_ranges_of_the_pair_search(hip_ctx, haplotypes, reads, chunk_rows):
    """pairTile2Kernel cuts the tiles of a matrix into the ranges of its work items: 64 columns are 136 tiles = 128 in two slices +
    8 in 32 slices, 100 columns 325 tiles = 256 + 69, ... — every item stages the columns its tiles touch, several chunks of rows,
    blocks of 46, 30 or 14 rows by LDS-direct loads at 8-byte-aligned (odd) row offsets.  Kept pairs and order: the oracle's."""
    rng = np.random.default_rng(7300 + haplotypes)
    cl = small_cases.make_cluster(rng, 1, [haplotypes], n_haps=haplotypes, n_reads=reads)
    batch = ClusterBatch.from_clusters([cl])
    dev = hip_ctx.upload(batch)
    g = [[p] for p in range(len(cl["paths"]))]
    mult = [p["source_count"] for p in cl["paths"]]
    dg = hip_ctx.groups(dev, [0], [g], False)
    if chunk_rows:
        os.environ["RPVG_HIP_PAIR_CHUNK_ROWS"] = str(chunk_rows)
    try:
        got = dg.bounded_pair_posteriors(np.asarray(mult), 1e-6)
    finally:
        os.environ.pop("RPVG_HIP_PAIR_CHUNK_ROWS", None)
    M, noise, cnts = np_oracle.grouped_matrix(cl["rows"], g)
    assert M.shape[1] == haplotypes
    sets, post = pyoracle.group_posteriors(M, noise, cnts, mult, 2, bounded=True, min_rel_lik=1e-6)
    assert got[0][0] == sets, (len(got[0][0]), len(sets))
    assert small_cases.rel_close(got[0][1], post, rel=1e-9, floor=1e-300)


# ---- row classes of the group matrices (count-1 products, mid counts, logarithms) ---------------------------

def _mixed_class_cluster(rng, n_reads):
    """A cluster whose rows cover every class of the matrix row order: read counts 1, 2..8, above 8, and noise
    probabilities below the product path's floor (2^-30)."""
    cl = small_cases.make_cluster(rng, 3, [8, 6, 7], n_haps=24, n_reads=n_reads)
    rows = []
    for i, (cnt, noise, groups) in enumerate(cl["rows"]):
        cnt = int(rng.choice([1, 1, 1, 1, 2, 3, 5, 8, 9, 40, 700]))
        if noise < 1.0 and rng.random() < 0.1:
            noise = 1e-12  # sub-floor noise: the row takes the logarithm path whatever its count
        rows.append((cnt, noise, groups))
    cl["rows"] = rows
    return cl


@pytest.mark.parametrize("n_reads", [300, 6000])
def test_row_classes_give_the_same_sums(hip_ctx, n_reads):
    """Matrices with fewer than 1024 rows have two classes, larger ones three; every consumer (member-list sums,
    conditionals, both searches) must agree with numpy / the oracle on rows of all classes."""
    rng = np.random.default_rng(900 + n_reads)
    clusters = [_mixed_class_cluster(rng, n_reads), _mixed_class_cluster(rng, n_reads // 2)]
    batch = ClusterBatch.from_clusters(clusters)
    assert n_reads < 1000 or min(len(cl["rows"]) for cl in clusters) >= 1024
    dev = hip_ctx.upload(batch)
    groups, mult = [], []
    for cl in clusters:
        g, m = np_oracle.source_groups(cl["paths"])
        groups.append(g)
        mult.append(m)
    dg = hip_ctx.groups(dev, [0, 1], groups, True)
    for m, cl in enumerate(clusters):
        M, noise, counts = np_oracle.grouped_matrix(cl["rows"], groups[m])
        M = np_oracle.add_noise_and_normalize(M, noise)[:, :-1]
        G = len(groups[m])
        pairs = [(a, b) for a in range(G) for b in range(a, G)]
        want = np.array([np_oracle.set_loglik(M, noise, counts, p, 2) for p in pairs])
        assert small_cases.rel_close(dg.loglik([m] * len(pairs), pairs, 2.0), want, rel=1e-11, floor=1e-9)
        cond = dg.conditionals([m], [[1]], 2, 2.0, [len(g) for g in groups])[0]
        want_c = np.array([np_oracle.set_loglik(M, noise, counts, (1, k), 2) for k in range(G)])
        assert small_cases.rel_close(cond, want_c, rel=1e-11, floor=1e-9)
    for table in (False, True, "tiles"):
        if table is True:
            os.environ["RPVG_HIP_TABLE_MIN_WORK"] = "0"
        if table != "tiles":
            os.environ["RPVG_HIP_PAIR_TILES"] = "0"
        try:
            got = dg.bounded_pair_posteriors(np.concatenate(mult), 1e-3)
        finally:
            os.environ.pop("RPVG_HIP_TABLE_MIN_WORK", None)
            os.environ.pop("RPVG_HIP_PAIR_TILES", None)
        for m, cl in enumerate(clusters):
            M, noise, counts = np_oracle.grouped_matrix(cl["rows"], groups[m])
            M = np_oracle.add_noise_and_normalize(M, noise)[:, :-1]
            sets, post = pyoracle.group_posteriors(M, noise, counts, mult[m], 2, bounded=True, min_rel_lik=1e-3)
            assert got[m][0] == sets
            assert small_cases.rel_close(got[m][1], post, rel=1e-9, floor=1e-300)


def test_bounded_search_with_a_positive_threshold(hip_ctx):
    """min_rel_likelihood > 1 makes the log threshold positive: the prefix-maximum form of the pruning rule does not
    hold then, and the kernels fall back to the reference's sequential rule (no table path, no wave scan)."""
    rng = np.random.default_rng(931)
    clusters = [small_cases.make_cluster(rng, 2, [6, 5], n_haps=20, n_reads=800),
                small_cases.make_cluster(rng, 3, [9, 7, 8], n_haps=40, n_reads=9000)]
    dev = hip_ctx.upload(ClusterBatch.from_clusters(clusters))
    groups, mult = [], []
    for cl in clusters:
        g, m = np_oracle.source_groups(cl["paths"])
        groups.append(g)
        mult.append(m)
    dg = hip_ctx.groups(dev, [0, 1], groups, True)
    got = dg.bounded_pair_posteriors(np.concatenate(mult), 1.5)
    for m, cl in enumerate(clusters):
        M, noise, counts = np_oracle.grouped_matrix(cl["rows"], groups[m])
        M = np_oracle.add_noise_and_normalize(M, noise)[:, :-1]
        sets, post = pyoracle.group_posteriors(M, noise, counts, mult[m], 2, bounded=True, min_rel_lik=1.5)
        assert got[m][0] == sets
        assert small_cases.rel_close(got[m][1], post, rel=1e-9, floor=1e-300)


@pytest.mark.parametrize("n_paths,n_reads", [(180, 500), (560, 2500), (1100, 700)])
def test_bounded_search_with_more_columns_than_the_lds_rows_hold(hip_ctx, n_paths, n_reads):
    """More columns than the kernel keeps rows of pair log-likelihoods for in LDS (128 for the small-matrix kernel,
    512 for the other): one first column at a time, rows in global scratch.  The table path is switched off so that the
    in-workgroup search takes the matrix."""
    rng = np.random.default_rng(940 + n_paths)
    cl = small_cases.make_cluster(rng, 1, [n_paths], n_haps=n_paths, n_reads=n_reads)
    dev = hip_ctx.upload(ClusterBatch.from_clusters([cl]))
    groups = [[p] for p in range(len(cl["paths"]))]
    mult = [p["source_count"] for p in cl["paths"]]
    dg = hip_ctx.groups(dev, [0], [groups], False)
    os.environ["RPVG_HIP_TABLE_MIN_WORK"] = "1e300"
    os.environ["RPVG_HIP_PAIR_TILES"] = "0"
    try:
        got = dg.bounded_pair_posteriors(np.array(mult), 1e-3)
        sequential = got
    finally:
        os.environ.pop("RPVG_HIP_TABLE_MIN_WORK", None)
        os.environ.pop("RPVG_HIP_PAIR_TILES", None)
    tiled = dg.bounded_pair_posteriors(np.array(mult), 1e-3)  # the default search: several passes of tiles over the staged rows
    assert tiled[0][0] == sequential[0][0]
    assert small_cases.rel_close(tiled[0][1], sequential[0][1], rel=1e-9, floor=1e-300)
    M, noise, counts = np_oracle.grouped_matrix(cl["rows"], groups)
    assert M.shape[1] == n_paths and (M.shape[0] <= 512) == (n_paths != 560)  # (1 100 columns: too wide for the tiles too)
    sets, post = pyoracle.group_posteriors(M, noise, counts, mult, 2, bounded=True, min_rel_lik=1e-3)
    assert got[0][0] == sets
    assert small_cases.rel_close(got[0][1], post, rel=1e-9, floor=1e-300)


def test_groups_with_a_path_outside_the_cluster_are_reported_by_the_first_consumer(hip_ctx):
    """rpvg_hip_groups_build returns with its kernels queued; the validity flag they set surfaces at the first use."""
    from rpvg_amd import hip
    clusters = small_cases.make_batch_clusters(951, n_clusters=2, with_empty=False)
    dev = hip_ctx.upload(ClusterBatch.from_clusters(clusters))
    bad_path = len(clusters[0]["paths"]) + 3
    dg = hip_ctx.groups(dev, [0], [[[0], [bad_path]]], False)
    with pytest.raises(hip.EngineError, match="outside its cluster"):
        dg.loglik([0], [[0]], 1.0)


@pytest.mark.parametrize("masks", ["2", "1"])
def test_groups_with_a_path_listed_twice_are_reported(hip_ctx, masks):
    """With the columns of a path as a word (the default: groupsBuildWordKernel) or the paths of a column as bit masks
    (RPVG_HIP_BUILD_MASKS=1: groupsBuildMaskKernel) a column is a set of paths."""
    from rpvg_amd import hip
    clusters = small_cases.make_batch_clusters(952, n_clusters=2, with_empty=False)
    dev = hip_ctx.upload(ClusterBatch.from_clusters(clusters))
    os.environ["RPVG_HIP_BUILD_MASKS"] = masks
    try:
        dg = hip_ctx.groups(dev, [0], [[[0], [1, 1]]], False)
        with pytest.raises(hip.EngineError, match="lists a path twice"):
            dg.loglik([0], [[0]], 1.0)
    finally:
        os.environ.pop("RPVG_HIP_BUILD_MASKS", None)


@pytest.mark.parametrize("normalise", [True, False])
def test_matrices_from_path_masks_equal_the_matrices_from_path_lists(hip_ctx, normalise):
    """groupsBuildWordKernel (the default up to 64 columns: a path's columns as one word, a wave per 64 rows, a lane per row, the
    cells in registers) and groupsBuildMaskKernel (RPVG_HIP_BUILD_MASKS=1, and wider matrices: a column's paths as a bit mask)
    add a row's entries in the order groupsBuildTileKernel / groupsBuildKernel (RPVG_HIP_BUILD_MASKS=0) do: the same values
    to the bit — seen through the log-likelihoods of every column and of pairs.  One word per column, two words (75 paths),
    more than 64 columns (two blocks), a cluster of one row, rows of more entries than the word kernel loads ahead."""
    rng = np.random.default_rng(961)
    clusters = small_cases.make_batch_clusters(962, n_clusters=8, with_empty=False)
    clusters.append(small_cases.make_cluster(rng, 3, [30, 25, 20], n_haps=100, n_reads=700))
    clusters.append(small_cases.make_cluster(rng, 1, [70], n_haps=70, n_reads=300))
    clusters.append(small_cases.make_cluster(rng, 2, [3, 2], n_haps=4, n_reads=1, empty_read_frac=0.0))
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    groups = []
    for cl in clusters:
        if normalise:
            g, _ = np_oracle.source_groups(cl["paths"])
        else:
            g = [[p] for p in range(len(cl["paths"]))]
        groups.append(g)
    assert max(len(g) for g in groups) > 64 and max(len(cl["paths"]) for cl in clusters) > 64
    entries = np.diff(batch.grp_idx_off.astype(np.int64)[batch.row_grp_off.astype(np.int64)])
    assert entries.max() > 4 and sum(1 for g in groups if len(g) <= 64) >= 3
    mats = list(range(len(clusters)))
    requests_m, requests_c = [], []
    for m, g in enumerate(groups):
        for c in range(len(g)):
            requests_m.append(m)
            requests_c.append([c, (c * 7 + 3) % len(g)])
    results = []
    for masks in ("2", "1", "0"):
        os.environ["RPVG_HIP_BUILD_MASKS"] = masks
        try:
            dg = hip_ctx.groups(dev, mats, groups, normalise)
            single = dg.loglik(requests_m, [[c[0]] for c in requests_c], 1.0)
            pairs = dg.loglik(requests_m, requests_c, 2.0)
        finally:
            os.environ.pop("RPVG_HIP_BUILD_MASKS", None)
        results.append((single, pairs))
    assert np.all(np.isfinite(results[0][0])) and np.all(np.isfinite(results[0][1]))
    for other in results[1:]:
        assert np.array_equal(results[0][0], other[0]) and np.array_equal(results[0][1], other[1])


def test_upload_from_counts_of_one_byte_equals_the_upload_from_offsets(hip_ctx):
    """rpvg_cluster_batch::row_grp_count8 / grp_idx_count8 (include/rpvg_batch.h): the groups of every row and the paths of every
    group as one byte each, summed up on the device behind the copy — the same batch as from the offsets (its matrices, through
    their log-likelihoods, and an EM solve); a count that does not fit sends the offsets; totals that the counts do not add up to
    are reported."""
    import ctypes as C
    from rpvg_amd import hip
    rng = np.random.default_rng(981)
    clusters = small_cases.make_batch_clusters(982, n_clusters=12, with_empty=True)
    clusters.append(small_cases.make_cluster(rng, 2, [9, 7], n_haps=12, n_reads=400))
    # (groups of several paths: paths that share a probability)
    clusters.append(dict(paths=[dict(group_id=p // 3, source_count=1, source_ids=[p % 4], effective_length=100.0) for p in range(6)],
                         rows=[(3, 0.05, [(0.125, [0, 4, 5]), (0.5, [1, 2])]), (1, 0.01, [(0.25, [3])]), (2, 0.5, [(0.0625, [0, 1, 2, 3, 4, 5])])]))
    batch = ClusterBatch.from_clusters(clusters)
    assert batch.counts8() is not None and int(np.diff(batch.grp_idx_off.astype(np.int64)).max()) > 1
    cb = batch.as_c(True)
    assert bool(cb.row_grp_count8) and cb.num_groups == len(batch.grp_prob) and cb.num_entries == len(batch.path_idx)
    results = []
    for compact in (True, False):
        dev = hip_ctx.upload(batch, compact=compact)
        try:
            mats = [k for k, cl in enumerate(clusters) if cl["rows"]]
            groups = [[[p] for p in range(len(clusters[k]["paths"]))] for k in mats]
            dg = hip_ctx.groups(dev, mats, groups, False)
            req_m = [m for m, g in enumerate(groups) for _ in g]
            req_c = [[c] for g in groups for c in range(len(g))]
            single = dg.loglik(req_m, req_c, 1.0)
            abund, noise, total, iters = hip_ctx.em_solve(dev, mats, [list(range(len(clusters[k]["paths"]))) for k in mats])
            results.append((single, np.concatenate(abund), noise, total, iters, dev.cluster_totals()))
        finally:
            dev.free()
    (single_a, abund_a, noise_a, total_a, _, totals_a), (single_b, abund_b, noise_b, total_b, _, totals_b) = results
    assert np.array_equal(single_a, single_b) and np.array_equal(total_a, total_b) and np.array_equal(totals_a, totals_b)
    # (the EM's column sums have one order of additions: em_sparse.hip, emSparseProblem)
    assert np.array_equal(abund_a, abund_b) and np.array_equal(noise_a, noise_b)

    # a group of 300 paths: the counts do not fit, the offsets travel
    wide = small_cases.make_batch_clusters(983, n_clusters=2, with_empty=False)
    n = 300
    wide[0]["paths"] = [dict(group_id=0, source_count=1, source_ids=[p % 7], effective_length=100.0) for p in range(n)]
    wide[0]["rows"] = [(1, 0.01, [(0.5, list(range(n)))])] + [(2, 0.02, [(0.25, [3, 5]), (0.5, [7])])]
    wide_batch = ClusterBatch.from_clusters(wide)
    assert wide_batch.counts8() is None and not bool(wide_batch.as_c(True).row_grp_count8)
    hip_ctx.upload(wide_batch, compact=True).free()

    # totals that are not the counts' sums
    cb = batch.as_c(True)
    cb.row_grp_off32 = None
    cb.grp_idx_off32 = None
    cb.num_entries = cb.num_entries - 1
    handle = C.c_void_p()
    assert hip.lib().rpvg_hip_batch_upload(hip_ctx.handle, C.byref(cb), C.byref(handle)) != 0
    assert "do not add up" in hip.lib().rpvg_hip_last_error().decode()
    cb.num_entries = cb.num_entries + 1
    assert hip.lib().rpvg_hip_batch_upload(hip_ctx.handle, C.byref(cb), C.byref(handle)) == 0  # (without any offsets)
    hip.lib().rpvg_hip_batch_free(hip_ctx.handle, handle)


@pytest.mark.parametrize("compact", [False, True], ids=["offsets", "counts"])
def test_upload_reports_the_first_row_that_breaks_an_invariant(hip_ctx, compact):
    """The rows of a batch are checked on the device, behind their copy (validateRowsKernel: over the offsets the caller wrote, or
    over the running sums of its counts of one byte); the host words the message."""
    from rpvg_amd import hip
    clusters = small_cases.make_batch_clusters(977, n_clusters=4, with_empty=False)
    good = ClusterBatch.from_clusters(clusters)
    hip_ctx.upload(good, compact=compact)  # nothing to report

    bad = ClusterBatch.from_clusters(clusters)
    last = len(bad.row_noise) - 1
    bad.row_noise[last] = 1.5
    bad.row_noise[last // 2] = 0.0
    with pytest.raises(hip.EngineError, match=r"row %d has noise probability 0 outside \(0, 1\]" % (last // 2)):
        hip_ctx.upload(bad, compact=compact)

    bad = ClusterBatch.from_clusters(clusters)
    row = int(bad.cluster_row_off[1]) + 2  # a row of the second cluster
    entry = int(bad.grp_idx_off[int(bad.row_grp_off[row])])
    n_paths = int(bad.cluster_path_off[2] - bad.cluster_path_off[1])
    bad.path_idx[entry] = n_paths
    bad.path_idx[int(bad.grp_idx_off[int(bad.row_grp_off[row + 3])])] = n_paths + 5  # (not the first)
    with pytest.raises(hip.EngineError, match=r"row %d refers to path %d of a cluster with %d paths" % (row, n_paths, n_paths)):
        hip_ctx.upload(bad, compact=compact)

    if not compact:  # (counts cannot contradict each other)
        bad = ClusterBatch.from_clusters(clusters)
        bad.grp_idx_off[1] = bad.grp_idx_off[-1] + 7
        with pytest.raises(hip.EngineError, match="inconsistent group or entry offsets"):
            hip_ctx.upload(bad)
    hip_ctx.upload(good, compact=compact)  # the context is still usable


def test_em_problem_sets_in_two_passes_when_the_storage_bound_is_over_budget(hip_ctx):
    """The compacted rows of the EM problems are stored by a bound (a problem keeps at most its cluster's rows and
    entries) so that one kernel counts and fills; over a budget (RPVG_HIP_EM_BOUND_BYTES) a counting pass comes first.
    Same solutions either way."""
    clusters = small_cases.make_batch_clusters(991, n_clusters=9, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    problems = []
    for k, cl in enumerate(clusters):
        n = len(cl["paths"])
        problems.append((k, list(range(n))))
        if n > 2:
            problems.append((k, list(range(0, n, 2))))
    owners, columns = [k for k, _ in problems], [c for _, c in problems]
    one_pass = hip_ctx.em_solve(dev, owners, columns, 10000, 1e-3)
    os.environ["RPVG_HIP_EM_BOUND_BYTES"] = "1"
    try:
        two_pass = hip_ctx.em_solve(dev, owners, columns, 10000, 1e-3)
    finally:
        os.environ.pop("RPVG_HIP_EM_BOUND_BYTES", None)
    assert np.array_equal(one_pass[3], two_pass[3]) and np.array_equal(one_pass[2], two_pass[2])  # iterations, totals
    assert small_cases.rel_close(one_pass[1], two_pass[1], rel=1e-12, floor=1e-12)
    for a, b in zip(one_pass[0], two_pass[0]):
        assert small_cases.rel_close(a, b, rel=1e-12, floor=1e-12)  # (LDS atomics: the last bits are run-dependent)


# ---- the Gibbs sampler of the group posteriors on the device (rpvg_hip_group_gibbs) ---------------------------------
def _mt19937_words(seed, skip, n):
    """Outputs skip .. skip + n of std::mt19937(seed) (numpy's legacy seeding is the reference implementation's init_genrand)."""
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)
    return [int(w) for w in bg.random_raw(skip + n)[skip:]]


def _gibbs_model(G, group_size, conditional, words):
    """estimatePathGroupPosteriorsGibbs (src/path_estimator.cpp:475-589) in plain Python over a list of generator outputs,
    with libstdc++'s distributions restated (GCC 11: bits/uniform_int_dist.h:246-270, bits/random.tcc:2656-2713,3348-3380).
    conditional(others) -> log-likelihood + log frequency of every candidate column.  Returns (sets in order of first
    appearance, counts, words taken)."""
    import bisect
    import math
    pos = 0

    def nxt():
        nonlocal pos
        pos += 1
        return words[pos - 1]

    def uniform_below(rng):
        product = nxt() * rng
        low = product & 0xffffffff
        if low < rng:
            threshold = ((1 << 32) - rng) % rng
            while low < threshold:
                product = nxt() * rng
                low = product & 0xffffffff
        return product >> 32

    def add_log(a, b):
        return a + math.log1p(math.exp(b - a)) if a > b else b + math.log1p(math.exp(a - b))

    chains = 10 + int(math.floor(0.01 * group_size * G + 0.5))
    burn = 50 + int(math.floor(0.025 * group_size * G + 0.5))
    its = 100 + int(math.floor(0.05 * group_size * G + 0.5))
    memo, order, counts = {}, [], {}
    for _ in range(chains):
        cur = [uniform_below(G) for _ in range(group_size)]
        for it in range(burn + its):
            for slot in range(group_size):
                others = tuple(cur[k] for k in range(group_size) if k != slot)
                if others not in memo:
                    ll = conditional(others)
                    total = -1.7976931348623157e308
                    for v in ll:
                        total = add_log(total, v)
                    p = [math.exp(v - total) for v in ll]
                    s = 0.0
                    for v in p:
                        s += v
                    cp, run = [], 0.0
                    for v in p:
                        run += v / s
                        cp.append(run)
                    cp[-1] = 1.0
                    memo[others] = cp
                if G >= 2:
                    first, second = nxt(), nxt()
                    u = (float(first) + float(second) * 4294967296.0) / 18446744073709551616.0
                    if u >= 1.0:
                        u = math.nextafter(1.0, 0.0)
                    cur[slot] = bisect.bisect_left(memo[others], u)
                else:
                    cur[slot] = 0  # a distribution of fewer than two weights draws nothing
            if it >= burn:
                key = tuple(sorted(cur))
                if key not in counts:
                    order.append(key)
                    counts[key] = 0
                counts[key] += 1
    return order, [counts[k] for k in order], pos, (chains, burn, its)


def _mt_temper(y):
    y ^= y >> 11
    y ^= (y << 7) & 0x9d2c5680
    y ^= (y << 15) & 0xefc60000
    y ^= y >> 18
    return y & 0xffffffff


@pytest.mark.parametrize("group_size", [1, 2])
def test_device_gibbs_sampler_follows_a_python_model_draw_for_draw(hip_ctx, group_size):
    """rpvg_hip_group_gibbs against a sequential Python model of the reference's sampler that draws from the same
    std::mt19937 words: same sets in the same order with the same counts, the same number of words taken from every
    generator (one generator serves two problems, one matrix has a single column), and state words that continue the
    stream.  The model takes its log-likelihoods from rpvg_hip_group_conditionals: this test is about the chains."""
    import math
    clusters = small_cases.make_batch_clusters(811, n_clusters=6, with_empty=False)
    clusters.append(small_cases.make_batch_clusters(812, n_clusters=1, with_empty=False)[0])
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    mats = list(range(len(clusters)))
    groups = [[[p] for p in range(len(cl["paths"]))] for cl in clusters]
    groups[-1] = [list(range(len(clusters[-1]["paths"])))]  # one column: every path of the cluster
    num_cols = [len(g) for g in groups]
    dg = hip_ctx.groups(dev, mats, groups, False)
    rng = np.random.default_rng(5)
    log_freq = [np.log(rng.integers(1, 5, size=G) / 7.0) for G in num_cols]
    generator_problems = [[0], [1, 2], [3], [4], [5], [6]]
    seeds = [(11, 0), (12, 100), (13, 623), (14, 624), (15, 7), (16, 1000)]

    def conditional_of(m):
        def conditional(others):
            got = dg.conditionals([m], [list(others)], group_size, float(group_size), num_cols)[0]
            return [float(x) + float(y) for x, y in zip(got, log_freq[m])]
        return conditional

    want, want_words, streams, shapes = {}, [], [], {}
    for problems, (seed, skip) in zip(generator_problems, seeds):
        words = _mt19937_words(seed, skip, 200000)
        taken = 0
        for m in problems:
            order, counts, used, shape = _gibbs_model(num_cols[m], group_size, conditional_of(m), words[taken:])
            want[m] = (order, counts)
            shapes[m] = shape
            taken += used
        want_words.append(taken)
        streams.append(words)
    got, words_consumed, state, (rounds, conditionals) = dg.gibbs(
        mats, group_size, [shapes[m][0] for m in mats], [shapes[m][1] for m in mats], [shapes[m][2] for m in mats], log_freq,
        generator_problems, [s[:624] for s in streams])
    assert [int(w) for w in words_consumed] == want_words
    for m in mats:
        assert got[m][0] == want[m][0], f"sets of problem {m}"
        assert got[m][1] == want[m][1], f"counts of problem {m}"
        assert sum(got[m][1]) == shapes[m][0] * shapes[m][2]
    assert got[6][0] == [(0,) * group_size]
    assert rounds >= 1 and conditionals >= sum(1 for G in num_cols if G >= 2)  # (one column: nothing to evaluate)
    for g, taken in enumerate(want_words):
        if taken >= 624:
            assert [_mt_temper(int(x)) for x in state[g]] == streams[g][taken - 624:taken]


def test_device_gibbs_sampler_reports_what_it_does_not_take(hip_ctx):
    """Group sizes above 2 are 'unsupported' (the caller's host-driven sampler takes them), a problem on a matrix that does
    not exist is an invalid argument, and a call without problems is an empty result."""
    from rpvg_amd import hip as hip_mod
    clusters = small_cases.make_batch_clusters(813, n_clusters=3, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    groups = [[[p] for p in range(len(cl["paths"]))] for cl in clusters]
    dg = hip_ctx.groups(dev, list(range(len(clusters))), groups, False)
    log_freq = [np.zeros(len(g)) for g in groups]
    words = [_mt19937_words(5, 0, 624)]
    with pytest.raises(hip_mod.EngineError) as unsupported:
        dg.gibbs([0], 3, [10], [50], [100], log_freq[:1], [[0]], words)
    assert "group size 3" in str(unsupported.value)
    with pytest.raises(hip_mod.EngineError) as invalid:
        dg.gibbs([7], 2, [10], [50], [100], log_freq[:1], [[0]], words)
    assert "matrix 7" in str(invalid.value)
    got, consumed, _, (rounds, conditionals) = dg.gibbs([], 2, [], [], [], [], [], np.zeros((0, 624), np.uint32))
    assert got == [] and len(consumed) == 0 and rounds == 0 and conditionals == 0


def test_spans_of_a_caller_that_never_reads_the_statistics_are_folded_on_the_way(hip_ctx):
    """Every kernel family of a call is bracketed by two HIP events until somebody reads the statistics (context.hip, spanBegin);
    a context that is never asked folds its finished spans every 1 024 without waiting for anything — their events do not pile up,
    and what they measured is in the statistics when somebody does ask."""
    clusters = small_cases.make_batch_clusters(995, n_clusters=3, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    try:
        mats = list(range(len(clusters)))
        columns = [list(range(len(cl["paths"]))) for cl in clusters]
        hip_ctx.reset_stats()
        first = hip_ctx.em_solve(dev, mats, columns)
        before = hip_ctx.stats()
        per_call = before["em_sparse_launches"]
        assert per_call > 0 and before["em_sparse_ms"] > 0
        calls = 400  # (several spans per call: the list passes 1 024 more than once)
        for _ in range(calls):
            last = hip_ctx.em_solve(dev, mats, columns)
        after = hip_ctx.stats()
        assert after["em_sparse_launches"] == per_call * (calls + 1)
        assert after["em_sparse_ms"] > before["em_sparse_ms"] * calls * 0.2
        assert np.array_equal(first[3], last[3])
    finally:
        dev.free()


def test_upload_from_the_narrow_forms_equals_the_upload_from_32_bits(hip_ctx):
    """rpvg_cluster_batch::path_idx16 / source_id16 / row_count8 + the list of the rows whose count does not fit a byte
    (include/rpvg_batch.h): widened on the device behind the copy — the same batch as from the 32-bit arrays: read totals, haplotype
    columns, log-likelihoods and EM solutions bit for bit; a cluster of 65 536 paths or an id of 65 536 keeps the 32-bit array."""
    from rpvg_amd import hip
    clusters = small_cases.make_batch_clusters(991, n_clusters=20, max_reads=600, with_empty=True)
    # read counts on both sides of a byte
    big = clusters[3]["rows"]
    clusters[3]["rows"] = [(c if i % 5 else 254 + (i % 4) * 300, z, g) for i, (c, z, g) in enumerate(big)]
    batch = ClusterBatch.from_clusters(clusters)
    forms = batch.narrow()
    assert set(forms) == {"path_idx16", "source_id16", "row_count8", "row_count_escape_row", "row_count_escape_count", "row_noise16", "row_noise_table"}
    assert len(forms["row_count_escape_row"]) > 2 and int(batch.row_count.max()) > 255
    cb = batch.as_c(True, True)
    assert bool(cb.path_idx16) and bool(cb.source_id16) and bool(cb.row_count8) and cb.num_row_count_escapes == len(forms["row_count_escape_row"])
    assert bool(cb.row_noise16) and cb.num_row_noise_values == len(np.unique(batch.row_noise))
    results = []
    for narrow in (True, False):
        dev = hip_ctx.upload(batch, compact=True, narrow=narrow)
        try:
            mats = [k for k, cl in enumerate(clusters) if cl["rows"]]
            groups = [[[p] for p in range(len(clusters[k]["paths"]))] for k in mats]
            dg = hip_ctx.groups(dev, mats, groups, False)
            req_m = [m for m, g in enumerate(groups) for _ in g]
            req_c = [[c] for g in groups for c in range(len(g))]
            single = dg.loglik(req_m, req_c, 1.0)
            abund, noise, total, iters = hip_ctx.em_solve(dev, mats, [list(range(len(clusters[k]["paths"]))) for k in mats])
            columns = [dev.source_columns(k) for k in range(batch.num_clusters)]
            results.append((single, np.concatenate(abund), noise, total, iters, dev.cluster_totals(), columns))
        finally:
            dev.free()
    a, b = results
    assert all(np.array_equal(x, y) for x, y in zip(a[:6], b[:6])) and a[6] == b[6]
    for k in range(batch.num_clusters):
        r0, r1 = int(batch.cluster_row_off[k]), int(batch.cluster_row_off[k + 1])
        assert a[5][k] == float(batch.row_count[r0:r1].astype(np.uint64).sum())

    # an index outside the table of noise values is an invalid noise probability
    forms["row_noise16"][5] = 60000
    try:
        with pytest.raises(hip.EngineError, match="noise probability"):
            hip_ctx.upload(batch, compact=True, narrow=True)
    finally:
        forms["row_noise16"][5] = np.searchsorted(forms["row_noise_table"], batch.row_noise[5])

    # an id of 65 536 and up: the source ids stay in 32 bits, the rest narrow
    fields = {name: getattr(batch, name).copy() for name in ClusterBatch._DTYPES}
    fields["source_id"] = fields["source_id"] + np.uint32(70000)
    far = ClusterBatch(**fields)
    assert "source_id16" not in far.narrow() and "path_idx16" in far.narrow()
    dev = hip_ctx.upload(far, compact=True, narrow=True)
    try:
        assert np.array_equal(dev.cluster_totals(), a[5])
        assert [dev.source_columns(k)[1] for k in range(far.num_clusters)] == [c[1] for c in a[6]]
    finally:
        dev.free()
