"""The path side of a batch on the device (rpvg_amd/csrc/path_sources.hip, subset_em.hip's merge kernel):
findPathSourceGroups (src/path_abundance_estimator.cpp:493-546) at upload, the group matrices from those columns, and the
posterior-weighted merge (src/path_abundance_estimator.cpp:702-749) — each against the host code it takes over (which the
model tests pin to the oracle) and the whole against the oracle."""
import numpy as np
import pytest

from oracle import pyoracle
from rpvg_amd import engine as eng_mod, hip
from rpvg_amd.batch import ClusterBatch, make_params
from tests import small_cases

pytestmark = pytest.mark.gpu


def host_columns(batch: ClusterBatch, k: int):
    """findPathSourceGroups as the host classes order it: haplotypes with the identical path list form a column, its
    multiplicity their number, columns by ascending smallest haplotype id."""
    p0, p1 = int(batch.cluster_path_off[k]), int(batch.cluster_path_off[k + 1])
    by_id = {}
    for p in range(p0, p1):
        for i in range(int(batch.path_source_off[p]), int(batch.path_source_off[p + 1])):
            by_id.setdefault(int(batch.source_id[i]), set()).add(p - p0)
    columns = {}
    for hap in sorted(by_id):
        columns.setdefault(tuple(sorted(by_id[hap])), []).append(hap)
    ordered = sorted(columns.items(), key=lambda kv: kv[1][0])
    return [len(haps) for _, haps in ordered], [list(paths) for paths, _ in ordered]


def with_sources(batch: ClusterBatch, source_lists):
    """The batch with other PathInfo::source_ids (one list per path)."""
    off = np.zeros(len(source_lists) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in source_lists])
    ids = np.array([i for x in source_lists for i in x], dtype=np.uint32)
    fields = {name: getattr(batch, name) for name in ClusterBatch._DTYPES}
    fields.update(path_source_off=off, source_id=ids if len(ids) else np.zeros(0, dtype=np.uint32),
                  path_source_count=np.array([max(1, len(x)) for x in source_lists], dtype=np.uint32))
    return ClusterBatch(**fields)


def test_source_columns_of_the_model_batches(hip_ctx):
    clusters = small_cases.make_batch_clusters(9101, n_clusters=60, with_empty=True)
    batch = ClusterBatch.from_clusters(clusters)
    dev = hip_ctx.upload(batch)
    try:
        assert dev.has_source_columns()
        for k in range(batch.num_clusters):
            assert dev.source_columns(k) == host_columns(batch, k), k
        totals = dev.cluster_totals()
        for k in range(batch.num_clusters):
            r0, r1 = int(batch.cluster_row_off[k]), int(batch.cluster_row_off[k + 1])
            assert totals[k] == float(batch.row_count[r0:r1].astype(np.uint64).sum())
    finally:
        dev.free()


@pytest.mark.parametrize("shape", ["wide_ids", "many_ids", "unordered_ids", "shared_lists"])
def test_source_columns_beyond_lds(hip_ctx, shape):
    """Clusters whose bit vectors or tables leave LDS (more than 1 024 ids in the range, more than 3 072 words), ids in any
    order within a path, and haplotypes that share their lists."""
    rng = np.random.default_rng(dict(wide_ids=11, many_ids=12, unordered_ids=13, shared_lists=14)[shape])
    clusters = small_cases.make_batch_clusters(9102, n_clusters=4, with_empty=False)
    n_paths = [len(c["paths"]) for c in clusters]
    batch = ClusterBatch.from_clusters(clusters)
    lists = []
    for k, n in enumerate(n_paths):
        if shape == "wide_ids":
            pool = rng.choice(200000, size=40, replace=False) + 1000 * k
        elif shape == "many_ids":
            pool = np.arange(5000) + 17
        elif shape == "unordered_ids":
            pool = rng.permutation(300)
        else:
            pool = np.arange(64)
        carried = {int(h): (rng.integers(0, n, size=max(1, n // 3)) if shape != "shared_lists" else rng.integers(0, n, size=2) % 2)
                   for h in pool}
        per_path = [[] for _ in range(n)]
        for h, paths in carried.items():
            for p in set(int(x) for x in paths):
                per_path[p].append(h)
        if shape == "unordered_ids":
            per_path = [list(rng.permutation(x)) for x in per_path]
        lists += per_path
    batch = with_sources(batch, lists)
    dev = hip_ctx.upload(batch)
    try:
        assert dev.has_source_columns()
        for k in range(batch.num_clusters):
            assert dev.source_columns(k) == host_columns(batch, k), (shape, k)
    finally:
        dev.free()


def test_source_columns_of_a_large_cluster(hip_ctx):
    """5 000 paths x 64 haplotypes (the shape of a configs[4] cluster): 79 words per haplotype, the arena in device memory."""
    rng = np.random.default_rng(77)
    clusters = [small_cases.make_cluster(rng, 2, (3, 2), n_haps=6, n_reads=5)]
    batch = ClusterBatch.from_clusters(clusters)
    n = 5000
    fields = {name: getattr(batch, name) for name in ClusterBatch._DTYPES}
    lists = [sorted(set(int(h) for h in rng.integers(0, 64, size=rng.integers(1, 6)))) for _ in range(n)]
    fields.update(cluster_path_off=np.array([0, n], dtype=np.uint64), path_group_id=np.arange(n, dtype=np.uint32) // 50,
                  path_effective_length=np.full(n, 1000.0))
    batch = with_sources(ClusterBatch(**fields), lists)
    dev = hip_ctx.upload(batch)
    try:
        assert dev.has_source_columns()
        assert dev.source_columns(0) == host_columns(batch, 0)
    finally:
        dev.free()


def test_id_ranges_too_wide_are_left_to_the_host(hip_ctx):
    """Haplotype ids spread over the 32-bit range: the device scratch does not hold a bit vector per id of the range; the
    batch has no columns and the estimator groups on the host (same estimates)."""
    clusters = small_cases.make_batch_clusters(9103, n_clusters=5, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    spread = batch.source_id.astype(np.uint64) * np.uint64(60000000) % np.uint64(4000000007)
    fields = {name: getattr(batch, name) for name in ClusterBatch._DTYPES}
    # (a multiplication by a constant modulo a prime keeps distinct ids distinct)
    fields.update(source_id=spread.astype(np.uint32))
    wide = ClusterBatch(**fields)
    dev = hip_ctx.upload(wide)
    try:
        assert not dev.has_source_columns()
    finally:
        dev.free()
    engine = eng_mod.Engine(0)
    try:
        got, _ = engine.run("haplotype-transcripts", make_params(), engine.prepare(wide))
        ref, _ = engine.run("haplotype-transcripts", make_params(), engine.prepare(batch))
    finally:
        engine.close()
    # the columns are ordered by smallest id, which the spreading permutes: the estimates are keyed by paths, not columns
    for g, r in zip(got, ref):
        gk, rk = g.keyed(), r.keyed()
        assert set(gk) == set(rk)
        for key in rk:
            assert small_cases.rel_close(gk[key][0], rk[key][0], rel=1e-9) and small_cases.rel_close(gk[key][1], rk[key][1], rel=1e-9)


def test_bad_source_offsets_are_reported(hip_ctx):
    clusters = small_cases.make_batch_clusters(9104, n_clusters=3, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    fields = {name: getattr(batch, name) for name in ClusterBatch._DTYPES}
    off = batch.path_source_off.copy()
    off[1] = off[2] + 3  # decreasing between the second and the third offset
    fields.update(path_source_off=off)
    with pytest.raises(hip.EngineError, match="path_source_off"):
        hip_ctx.upload(ClusterBatch(**fields))


def _flat_posterior_clusters(seed, n=3):
    rng = np.random.default_rng(seed)
    return ([small_cases.make_cluster(rng, 3, (12, 10, 8), n_haps=30, n_reads=int(rng.integers(3, 7)), empty_read_frac=0.0) for _ in range(n)] +
            [small_cases.make_cluster(rng, 1, (36,), n_haps=36, n_reads=2, empty_read_frac=0.0)])


def test_device_columns_and_merge_equal_the_host_path(monkeypatch):
    """`-i haplotype-transcripts` with the haplotype columns formed and the solutions merged on the device against the same
    call with both on the host (RPVG_AMD_HOST_SOURCE_GROUPS=1: findPathSourceGroups and the weighted merge of
    rpvg_amd/host/path_abundance_estimator.cpp): the identical sets, posteriors, abundances and noise counts, bit for bit —
    the additions are the same, in the same order, without fused multiply-adds — and the same EM problems."""
    clusters = small_cases.make_batch_clusters(9105, n_clusters=50, with_empty=True) + _flat_posterior_clusters(9106)
    batch = ClusterBatch.from_clusters(clusters)
    engine = eng_mod.Engine(0)
    try:
        prep = engine.prepare(batch)
        params = make_params()
        device, _ = engine.run("haplotype-transcripts", params, prep)
        monkeypatch.setenv("RPVG_AMD_HOST_SOURCE_GROUPS", "1")
        host, _ = engine.run("haplotype-transcripts", params, prep)
        monkeypatch.delenv("RPVG_AMD_HOST_SOURCE_GROUPS")
    finally:
        engine.close()
    most_sets = 0
    for k, (d, h) in enumerate(zip(device, host)):
        assert [list(s) for s in d.path_group_sets] == [list(s) for s in h.path_group_sets], k
        assert np.array_equal(d.posteriors, h.posteriors), k
        assert np.array_equal(d.abundances, h.abundances), k
        assert d.noise_count == h.noise_count and d.total_count == h.total_count, k
        assert list(d.em_cols) == list(h.em_cols) and list(d.em_iters) == list(h.em_iters), k
        most_sets = max(most_sets, len(d.path_group_sets))
    assert most_sets > 30
    ref, _ = pyoracle.run("haplotype-transcripts", make_params(), batch, 4)
    for g, r in zip(device, ref):
        gk, rk = g.keyed(), r.keyed()
        assert set(gk) == set(rk)
        for key in rk:
            assert small_cases.rel_close(gk[key][0], rk[key][0], rel=1e-4) and small_cases.rel_close(gk[key][1], rk[key][1], rel=1e-4)


@pytest.mark.parametrize("steps", ["begin_finish", "begin_queue_wait"])
def test_an_upload_in_steps_equals_the_upload_in_one_call(hip_ctx, steps):
    """rpvg_hip_batch_upload_begin (the copies) + _finish (the kernels behind them, any context of the device), and the second half
    itself as _finish_queue (no wait: a pipeline's uploader goes on copying) + _finish_wait (from any thread): the same batch as
    rpvg_hip_batch_upload makes — read totals, haplotype columns, an EM solve — and the same message for a row that breaks an
    invariant."""
    import ctypes as C
    clusters = small_cases.make_batch_clusters(9107, n_clusters=20, with_empty=True)
    batch = ClusterBatch.from_clusters(clusters)
    whole = hip_ctx.upload(batch)
    other = hip.Context(0)
    try:
        cb = batch.as_c(True)
        handle = C.c_void_p()
        hip._check(hip.lib().rpvg_hip_batch_upload_begin(other.handle, C.byref(cb), C.byref(handle)), "rpvg_hip_batch_upload_begin")
        if steps == "begin_finish":
            hip._check(hip.lib().rpvg_hip_batch_upload_finish(hip_ctx.handle, handle, C.byref(cb)), "rpvg_hip_batch_upload_finish")
        else:
            hip._check(hip.lib().rpvg_hip_batch_upload_finish_queue(other.handle, handle, C.byref(cb)), "rpvg_hip_batch_upload_finish_queue")
            hip._check(hip.lib().rpvg_hip_batch_upload_finish_wait(handle, C.byref(cb)), "rpvg_hip_batch_upload_finish_wait")
        stepped = hip.DeviceBatch.__new__(hip.DeviceBatch)
        stepped.ctx, stepped.host, stepped.handle = hip_ctx, batch, handle
        try:
            assert stepped.has_source_columns() and np.array_equal(stepped.cluster_totals(), whole.cluster_totals())
            for k in range(batch.num_clusters):
                assert stepped.source_columns(k) == whole.source_columns(k), k
            mats = [k for k, cl in enumerate(clusters) if cl["rows"]]
            columns = [list(range(len(clusters[k]["paths"]))) for k in mats]
            a = hip_ctx.em_solve(stepped, mats, columns)
            b = hip_ctx.em_solve(whole, mats, columns)
            assert np.allclose(np.concatenate(a[0]), np.concatenate(b[0]), rtol=1e-9, atol=1e-12) and np.array_equal(a[3], b[3])
        finally:
            stepped.free()
        bad = ClusterBatch.from_clusters(clusters)
        bad.row_noise[3] = 0.0
        cb = bad.as_c(True)
        hip._check(hip.lib().rpvg_hip_batch_upload_begin(other.handle, C.byref(cb), C.byref(handle)), "rpvg_hip_batch_upload_begin")
        if steps == "begin_finish":
            rc = hip.lib().rpvg_hip_batch_upload_finish(hip_ctx.handle, handle, C.byref(cb))
        else:
            hip._check(hip.lib().rpvg_hip_batch_upload_finish_queue(other.handle, handle, C.byref(cb)), "rpvg_hip_batch_upload_finish_queue")
            rc = hip.lib().rpvg_hip_batch_upload_finish_wait(handle, C.byref(cb))
        assert rc != 0 and "row 3 has noise probability 0" in hip.lib().rpvg_hip_last_error().decode()  # (the batch is freed by the call)
    finally:
        whole.free()
        other.close()

