"""A batch from its callers' segments (rpvg_hip_batch_upload_segments, include/rpvg_batch.h rpvg_cluster_segment): what
PathEstimator::estimate()'s call combiner hands to the GPU — one page-locked segment per cluster of the reference's loop
(src/main.cpp:829,976-977), joined by one kernel that reads them where they lie — against the upload of the joined batch:
the same read totals, haplotype columns and EM solutions, bit for bit; and the validation of what the kernel reads."""
import ctypes as C

import numpy as np
import pytest

from rpvg_amd import hip
from rpvg_amd.batch import ClusterBatch
from tests import small_cases

pytestmark = pytest.mark.gpu


def solve_all(ctx, dev, batch):
    ks = [k for k in range(batch.num_clusters) if batch.cluster_row_off[k + 1] > batch.cluster_row_off[k]]
    cols = [list(range(int(batch.cluster_path_off[k + 1] - batch.cluster_path_off[k]))) for k in ks]
    return ctx.em_solve(dev, ks, cols)


@pytest.mark.parametrize("with_paths", [True, False])
def test_segments_equal_the_joined_upload(hip_ctx, with_paths):
    clusters = small_cases.make_batch_clusters(9301, n_clusters=40, max_reads=900, with_empty=True)
    batch = ClusterBatch.from_clusters(clusters)
    segments = hip.PinnedSegments(batch, with_paths=with_paths)
    joined = hip_ctx.upload(batch)
    pulled = hip_ctx.upload_segments(batch, segments)
    try:
        assert pulled.has_source_columns() == with_paths
        assert np.array_equal(pulled.cluster_totals(), joined.cluster_totals())
        if with_paths:
            for k in range(batch.num_clusters):
                assert pulled.source_columns(k) == joined.source_columns(k), k
        a, b = solve_all(hip_ctx, pulled, batch), solve_all(hip_ctx, joined, batch)
        assert np.array_equal(a[3], b[3])  # iterations
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])  # (the EM's sums have one order of additions: em_sparse.hip)
        for x, y in zip(a[0], b[0]):
            assert np.array_equal(x, y)
    finally:
        pulled.free()
        joined.free()
        segments.free()


def test_segments_with_the_callers_columns_equal_the_device_columns(hip_ctx):
    """rpvg_cluster_segment::has_columns: the haplotype columns formed by the caller (findPathSourceGroups on the calling thread,
    as the reference does) travel with the segment instead of the source ids — the batch holds the columns the device would have
    formed, and an inconsistent column list or a path outside the cluster is refused."""
    from tests.test_hip_path_sources import host_columns
    clusters = small_cases.make_batch_clusters(9303, n_clusters=30, max_reads=500, with_empty=True)
    batch = ClusterBatch.from_clusters(clusters)
    columns = [host_columns(batch, k) for k in range(batch.num_clusters)]
    segments = hip.PinnedSegments(batch, columns=columns)
    joined = hip_ctx.upload(batch)
    pulled = hip_ctx.upload_segments(batch, segments)
    try:
        assert pulled.has_source_columns()
        assert np.array_equal(pulled.cluster_totals(), joined.cluster_totals())
        for k in range(batch.num_clusters):
            assert pulled.source_columns(k) == joined.source_columns(k) == columns[k], k
        a, b = solve_all(hip_ctx, pulled, batch), solve_all(hip_ctx, joined, batch)
        assert np.array_equal(a[3], b[3]) and all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
    finally:
        pulled.free()
        joined.free()
    k = next(k for k in range(batch.num_clusters) if len(columns[k][0]) > 1)
    for fault, expect in (("end", "inconsistent"), ("path", "refers to a path outside its cluster")):
        views = segments.arrays[k]
        saved = {name: views[name].copy() for name in ("col_end", "col_path")}
        if fault == "end":
            views["col_end"][0] = views["col_end"][-1] + 5
        else:
            views["col_path"][0] = 10 ** 6
        with pytest.raises(hip.EngineError) as err:
            hip_ctx.upload_segments(batch, segments)
        assert f"cluster {k} of the batch" in str(err.value) and expect in str(err.value), str(err.value)
        for name in saved:
            views[name][:] = saved[name]
    segments.free()


def test_a_large_cluster_takes_several_slices(hip_ctx):
    """One cluster of tens of thousands of rows next to small ones: the kernel's slices walk it."""
    from rpvg_amd import synth
    batch = synth.generate(seed=77, num_clusters=12, total_paths=300, total_reads=400000)
    segments = hip.PinnedSegments(batch)
    joined = hip_ctx.upload(batch)
    pulled = hip_ctx.upload_segments(batch, segments)
    try:
        assert int(np.diff(batch.cluster_row_off).max()) > 4096
        assert np.array_equal(pulled.cluster_totals(), joined.cluster_totals())
        for k in range(batch.num_clusters):
            assert pulled.source_columns(k) == joined.source_columns(k), k
        a, b = solve_all(hip_ctx, pulled, batch), solve_all(hip_ctx, joined, batch)
        assert np.array_equal(a[3], b[3])
        for x, y in zip(a[0], b[0]):
            assert np.array_equal(x, y)
    finally:
        pulled.free()
        joined.free()
        segments.free()


@pytest.mark.parametrize("fault", ["noise", "path", "row_offsets", "group_offsets", "source_offsets", "foreign_block", "outside_block"])
def test_invalid_segments_are_refused(hip_ctx, fault):
    clusters = small_cases.make_batch_clusters(9302, n_clusters=8, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    segments = hip.PinnedSegments(batch)
    keep = None
    k = 5
    views = segments.arrays[k]
    if fault == "noise":
        views["row_noise"][3] = 1.5
        expect = "cluster 5 of the batch: a row has a noise probability"
    elif fault == "path":
        views["path_idx"][0] = 10000
        expect = "cluster 5 of the batch: a row or a haplotype column refers to a path outside its cluster"
    elif fault == "row_offsets":
        views["row_grp_off"][2] = views["row_grp_off"][-1] + 7
        expect = "cluster 5 of the batch: inconsistent"
    elif fault == "group_offsets":
        views["grp_idx_off"][-1] += 1
        expect = "cluster 5 of the batch: inconsistent"
    elif fault == "source_offsets":
        views["path_source_off"][1] = views["path_source_off"][-1] + 3
        expect = "cluster 5 of the batch: inconsistent"
    elif fault == "foreign_block":
        keep = np.zeros(int(segments.segments[k].bytes) + 64, dtype=np.uint8)
        segments.segments[k].base = keep.ctypes.data
        expect = "segment 5 does not lie in a block of rpvg_hip_pinned_alloc"
    else:
        segments.segments[k].row_noise_at = segments.segments[k].bytes
        expect = "an array of segment 5 is misaligned or outside its block"
    try:
        with pytest.raises(hip.EngineError) as err:
            hip_ctx.upload_segments(batch, segments)
        assert expect in str(err.value), str(err.value)
        # the context is usable afterwards
        good = hip.PinnedSegments(batch)
        dev = hip_ctx.upload_segments(batch, good)
        dev.free()
        good.free()
    finally:
        segments.free()
