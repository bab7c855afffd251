"""Large clusters behind the estimator classes: a cluster too big for one workgroup goes through the same
PathAbundanceEstimator::estimate path (src/path_abundance_estimator.cpp:18-45, called per cluster at src/main.cpp:977) as
any other, and the size bin of its EM problem sends it to the whole-GPU route of rpvg_hip_em_solve (rpvg_amd/csrc/em_grid.hip):
CSR route (thread per row / wavefront per row) or dense route.  Compared with the CPU oracle: EM iteration counts exact,
abundances to 1e-6 relative.
"""
import numpy as np
import pytest

from oracle import pyoracle
from rpvg_amd import engine as eng_mod, hip
from rpvg_amd.batch import make_params
from tests import large_cases, small_cases

pytestmark = pytest.mark.gpu

REL = 1e-6
GRID = "emGridAccumKernel"


@pytest.fixture(scope="module")
def engine():
    e = eng_mod.Engine(0)
    yield e
    e.close()


def _compare(got, ref):
    assert len(got) == len(ref)
    for k, (g, r) in enumerate(zip(got, ref)):
        gk, rk = g.keyed(), r.keyed()
        assert set(gk) == set(rk), f"cluster {k}: group sets differ"
        for key, (post, ab) in rk.items():
            assert small_cases.rel_close(gk[key][0], post, rel=REL, floor=1e-8), (k, key)
            assert small_cases.rel_close(gk[key][1], ab, rel=REL), (k, key, gk[key][1], ab)
        assert g.total_count == r.total_count
        assert abs(g.noise_count - r.noise_count) <= REL * max(1.0, r.total_count)
        assert dict(zip(g.em_cols, g.em_iters)) == dict(zip(r.em_cols, r.em_iters)), f"cluster {k}: EM iterations differ"
        if r.total_count > 0:
            assert abs(g.abundances.sum() + g.noise_count - g.total_count) <= 1e-9 * g.total_count


def _run(engine, model, params, batch):
    engine.reset_stats()
    prep = engine.prepare(batch)
    try:
        got, _ = engine.run(model, params, prep)
    finally:
        prep.free()
    return got, engine.stats()


def test_sparse_cluster_takes_the_grid_route_and_matches_the_oracle(engine, monkeypatch):
    """60 000 rows x 150 paths, three paths per row, 1 % of the rows without a path (the scalar Z): thread-per-row CSR route."""
    monkeypatch.setenv("RPVG_HIP_EM_GRID_MIN_WORK", "100000")
    batch = large_cases.cluster_batch(60000, 150, 3, seed=1, noise_only_frac=0.01)
    params = make_params()
    got, stats = _run(engine, "transcripts", params, batch)
    ref, _ = pyoracle.run("transcripts", params, batch, 1)
    _compare(got, ref)
    assert stats["em_kernel"][GRID]["problems"] == 1
    assert stats["em_kernel"][GRID]["iterations"] == ref[0].em_iters[0]
    assert stats["em_dense_launches"] == 0


def test_the_same_cluster_below_the_threshold_stays_on_one_workgroup(engine, monkeypatch):
    monkeypatch.setenv("RPVG_HIP_EM_GRID_MIN_WORK", "0")
    batch = large_cases.cluster_batch(60000, 150, 3, seed=1, noise_only_frac=0.01)
    params = make_params()
    got, stats = _run(engine, "transcripts", params, batch)
    ref, _ = pyoracle.run("transcripts", params, batch, 1)
    _compare(got, ref)
    assert stats["em_kernel"][GRID]["problems"] == 0


def test_200000_row_cluster_through_the_estimator_class(engine):
    """a single 200 000 x 400 cluster behind `-i transcripts` with the default threshold.  The oracle is
    dense (1.3 s per iteration at this size): a budget of 15 iterations here, convergence in the other tests."""
    batch = large_cases.cluster_batch(200000, 400, 3, seed=2)
    params = make_params(max_em_its=15)
    got, stats = _run(engine, "transcripts", params, batch)
    ref, _ = pyoracle.run("transcripts", params, batch, 1)
    _compare(got, ref)
    assert stats["em_kernel"][GRID]["problems"] == 1 and stats["em_kernel"][GRID]["iterations"] == 15


def test_long_rows_take_the_wavefront_per_row_route(engine):
    """8 000 rows with 40 of 400 paths each (320 000 entries: over the default threshold, fill 10 %)."""
    batch = large_cases.cluster_batch(8000, 400, 40, seed=4)
    params = make_params()
    got, stats = _run(engine, "transcripts", params, batch)
    ref, _ = pyoracle.run("transcripts", params, batch, 1)
    _compare(got, ref)
    assert stats["em_kernel"][GRID]["problems"] == 1 and stats["em_dense_launches"] == 0


@pytest.mark.parametrize("rows,paths", [(3000, 200), (1500, 300)])
def test_dense_cluster_takes_the_dense_route(engine, rows, paths):
    """Every row touches every path: the dense copy is the smaller representation — em_dense.hip's streaming kernels behind
    rpvg_hip_em_solve (narrow: a row per wave; wide, > 256 columns: a row per workgroup)."""
    batch = large_cases.cluster_batch(rows, paths, paths, seed=3, noise_only_frac=0.02)
    params = make_params()
    got, stats = _run(engine, "transcripts", params, batch)
    ref, _ = pyoracle.run("transcripts", params, batch, 1)
    _compare(got, ref)
    assert stats["em_dense_launches"] == ref[0].em_iters[0]
    assert stats["em_kernel"][GRID]["problems"] == 1 and stats["em_kernel"][GRID]["iterations"] == 0  # (accounted as dense)


def test_dense_route_at_50000_rows_by_2000_paths_matches_the_oracle(engine):
    """The shape of BASELINE.json configs[1] at a twentieth of its rows — 50 000 rows x 2 000 paths, every row touching every
    path: 10^8 entries — through PathAbundanceEstimator::estimateBatch: compaction, dense copy (emGridDenseBuildKernel), the wide
    streaming kernel (a row per workgroup) — against the oracle after a fixed budget of iterations (full convergence takes
    thousands; the oracle walks the dense matrix on one core)."""
    batch = large_cases.cluster_batch(50000, 2000, 2000, seed=11, noise_only_frac=0.01)
    params = make_params(max_em_its=10)
    got, stats = _run(engine, "transcripts", params, batch)
    ref, _ = pyoracle.run("transcripts", params, batch, 1)
    _compare(got, ref)
    assert ref[0].em_iters[0] == 10 and stats["em_dense_launches"] == 10
    assert stats["em_kernel"][GRID]["problems"] == 1 and stats["em_kernel"][GRID]["iterations"] == 0  # (accounted as dense)


def test_fused_dense_build_equals_the_csr_and_the_copy(monkeypatch):
    """A problem of the dense route gets its matrix from the compaction itself (em_sparse.hip, the fused build; RPVG_HIP_NO_FUSED_DENSE=1:
    compacted CSR, zeroed matrix, copy): the same matrix, so the same abundances bit for bit — for a problem over every path of its
    cluster (the `transcripts` model: no column map, no reading of path indices to count), for one over a part of the paths (rows
    that touch none of them drop out, the others keep only their selected entries) and for several in one solve."""
    from rpvg_amd.batch import ClusterBatch
    rng = np.random.default_rng(3)
    parts = [large_cases.cluster_batch(3000, 300, 300, seed=21, noise_only_frac=0.02), large_cases.cluster_batch(4000, 400, 300, seed=22, noise_only_frac=0.01),
             ClusterBatch.from_clusters(small_cases.make_batch_clusters(78, n_clusters=6))]
    batch = ClusterBatch.concat(parts)
    all0, all1 = list(range(300)), list(range(400))
    part0 = sorted(int(x) for x in rng.choice(300, size=170, replace=False))
    part1 = sorted(int(x) for x in rng.choice(400, size=220, replace=False))
    part2 = sorted(int(x) for x in rng.choice(300, size=200, replace=False))
    part3 = sorted(int(x) for x in rng.choice(400, size=260, replace=False))
    # (six problems of the dense route: four get their matrices from the compaction, the other two the two-step way, in one solve)
    problems = [(0, all0), (0, part0), (1, all1), (1, part1), (0, part2), (1, part3), (2, list(range(int(batch.cluster_path_off[3] - batch.cluster_path_off[2]))))]
    ctx = hip.Context(0)
    dev = ctx.upload(batch)
    try:
        out = {}
        for fused in (True, False):
            if fused:
                monkeypatch.delenv("RPVG_HIP_NO_FUSED_DENSE", raising=False)
            else:
                monkeypatch.setenv("RPVG_HIP_NO_FUSED_DENSE", "1")
            ctx.reset_stats()
            out[fused] = ctx.em_solve(dev, [k for k, _ in problems], [c for _, c in problems], max_em_its=25)
            out[fused] += (ctx.stats(),)
        a, b = out[True], out[False]
        assert a[4]["em_dense_launches"] == b[4]["em_dense_launches"] == 150     # six problems on the dense route, 25 iterations each
        assert b[4]["build_launches"] - a[4]["build_launches"] == 3             # ... six copy kernels against two and the launch that writes four matrices
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        for x, y in zip(a[0], b[0]):
            assert np.array_equal(x, y)
        assert all(x.sum() > 0 for x in a[0])
    finally:
        dev.free()
        ctx.close()


def test_large_and_small_clusters_in_one_batch(engine, monkeypatch):
    """The grid bin next to the one-workgroup bins of the same solve."""
    monkeypatch.setenv("RPVG_HIP_EM_GRID_MIN_WORK", "50000")
    from rpvg_amd.batch import ClusterBatch
    small = ClusterBatch.from_clusters(small_cases.make_batch_clusters(77, n_clusters=12))
    big = [large_cases.cluster_batch(30000, 90, 3, seed=5), large_cases.cluster_batch(2000, 60, 60, seed=6)]
    batch = ClusterBatch.concat([small] + big)
    params = make_params()
    got, stats = _run(engine, "transcripts", params, batch)
    ref, _ = pyoracle.run("transcripts", params, batch, 2)
    _compare(got, ref)
    assert stats["em_kernel"][GRID]["problems"] == 2


def test_strains_on_a_large_cluster_collapses_rows_before_the_grid_em(engine, monkeypatch):
    """`-i strains`: minimum path cover, then readCollapseProbabilityMatrix + EM on the cover (src/path_abundance_estimator.cpp:
    217-295) — the grid route reads the merged read counts of a problem the collapse merged rows in."""
    monkeypatch.setenv("RPVG_HIP_EM_GRID_MIN_WORK", "40000")
    batch = large_cases.cluster_batch(25000, 12, 2, seed=8, max_count=2)
    params = make_params()
    got, stats = _run(engine, "strains", params, batch)
    ref, _ = pyoracle.run("strains", params, batch, 1)
    _compare(got, ref)
    assert stats["em_kernel"][GRID]["problems"] == 1


def test_haplotype_transcripts_on_a_large_cluster(engine, monkeypatch):
    """The nested model builds its EM problems on the device (rpvg_hip_nested_subset_em): the subsets of a large cluster land
    in the grid bin there too."""
    monkeypatch.setenv("RPVG_HIP_EM_GRID_MIN_WORK", "5000")
    batch = large_cases.cluster_batch(30000, 24, 2, seed=9, groups=3, haplotypes=6)
    params = make_params()
    got, stats = _run(engine, "haplotype-transcripts", params, batch)
    ref, _ = pyoracle.run("haplotype-transcripts", params, batch, 1)
    _compare(got, ref)
    assert stats["em_kernel"][GRID]["problems"] >= 1
