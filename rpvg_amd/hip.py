"""ctypes binding of ``include/rpvg_hip.h`` (librpvg_hip.so) for tests and bench.

Thin: every function maps 1:1 to a C-ABI entry point.  There is no fallback:
if the shared library is missing or no GPU is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .batch import CClusterBatch, ClusterBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librpvg_hip.so")

# every symbol include/rpvg_hip.h declares
EXPORTS = [
    "rpvg_hip_device_count", "rpvg_hip_create", "rpvg_hip_create_uploader", "rpvg_hip_destroy", "rpvg_hip_last_error", "rpvg_hip_synchronize",
    "rpvg_hip_device_info", "rpvg_hip_malloc", "rpvg_hip_free", "rpvg_hip_memcpy_h2d", "rpvg_hip_memcpy_d2h",
    "rpvg_hip_batch_upload", "rpvg_hip_batch_free", "rpvg_hip_em_solve", "rpvg_hip_em_dense",
    "rpvg_hip_dense_from_cluster", "rpvg_hip_groups_build", "rpvg_hip_groups_free", "rpvg_hip_groups_collapse_info", "rpvg_hip_group_loglik",
    "rpvg_hip_synth_dense_cluster", "rpvg_hip_stats_get", "rpvg_hip_stats_reset", "rpvg_hip_stats_intervals", "rpvg_hip_em_kernel_name",
    "rpvg_hip_gibbs_read_counts", "rpvg_hip_min_path_cover", "rpvg_hip_bounded_pair_posteriors", "rpvg_hip_pair_posteriors_get", "rpvg_hip_pair_posteriors_free",
    "rpvg_hip_em_dense_sharded", "rpvg_hip_synth_dense_rows", "rpvg_hip_synth_dense_cluster_batch", "rpvg_hip_comm_unique_id", "rpvg_hip_comm_init",
    "rpvg_hip_comm_destroy", "rpvg_hip_comm_allreduce_sum_f64", "rpvg_hip_comm_init_all", "rpvg_hip_gather", "rpvg_hip_host_register", "rpvg_hip_host_unregister", "rpvg_hip_group_conditionals",
    "rpvg_hip_group_gibbs", "rpvg_hip_gibbs_sets_get", "rpvg_hip_gibbs_sets_free",
    "rpvg_hip_alignments_upload", "rpvg_hip_alignments_free", "rpvg_hip_read_rows_build", "rpvg_hip_read_rows_to_batch",
    "rpvg_hip_read_rows_view", "rpvg_hip_read_rows_sizes", "rpvg_hip_read_rows_free", "rpvg_hip_path_clusters", "rpvg_hip_debug_log",
    "rpvg_hip_nested_subset_em", "rpvg_hip_subset_em_get", "rpvg_hip_subset_em_free",
    "rpvg_hip_batch_cluster_totals", "rpvg_hip_batch_has_source_columns", "rpvg_hip_batch_source_columns_sizes",
    "rpvg_hip_batch_source_columns_get", "rpvg_hip_groups_build_from_sources", "rpvg_hip_groups_build_single_paths", "rpvg_hip_batch_upload_begin", "rpvg_hip_batch_upload_finish", "rpvg_hip_create_with_streams",
    "rpvg_hip_batch_upload_finish_queue", "rpvg_hip_batch_upload_finish_wait",
    "rpvg_hip_batch_upload_segments", "rpvg_hip_pinned_alloc", "rpvg_hip_pinned_free", "rpvg_hip_thread_wait_spin_us",
]

COMM_ID_BYTES = 128  # RPVG_HIP_COMM_ID_BYTES


class EngineError(RuntimeError):
    pass


class CEmProblems(C.Structure):
    _fields_ = [("num_problems", C.c_uint32), ("cluster", C.c_void_p), ("col_off", C.c_void_p), ("col_path", C.c_void_p),
                ("collapse_precision", C.c_double)]


class CEmResults(C.Structure):
    _fields_ = [("abundances", C.c_void_p), ("noise_count", C.c_void_p), ("total_count", C.c_void_p),
                ("iterations", C.c_void_p)]


class CGroupSpec(C.Structure):
    _fields_ = [("num_matrices", C.c_uint32), ("cluster", C.c_void_p), ("group_off", C.c_void_p),
                ("group_path_off", C.c_void_p), ("group_path", C.c_void_p), ("normalise", C.c_int32),
                ("collapse_precision", C.c_double)]


class CGibbsSpec(C.Structure):
    _fields_ = [("num_problems", C.c_uint32), ("group_size", C.c_uint32), ("matrix", C.c_void_p), ("num_chains", C.c_void_p),
                ("num_burn_its", C.c_void_p), ("num_gibbs_its", C.c_void_p), ("log_freq", C.c_void_p), ("num_generators", C.c_uint32),
                ("generator_problem_off", C.c_void_p), ("generator_problem", C.c_void_p), ("generator_words", C.c_void_p)]


class CGibbsSetsView(C.Structure):
    _fields_ = [("num_problems", C.c_uint32), ("group_size", C.c_uint32), ("set_off", C.POINTER(C.c_uint64)),
                ("first", C.POINTER(C.c_uint32)), ("second", C.POINTER(C.c_uint32)), ("count", C.POINTER(C.c_uint32)),
                ("words_consumed", C.POINTER(C.c_uint64)), ("generator_state", C.POINTER(C.c_uint32)), ("rounds", C.c_uint32),
                ("conditionals", C.c_uint64)]


class CPairPosteriorsView(C.Structure):
    _fields_ = [("num_matrices", C.c_uint32), ("pair_off", C.POINTER(C.c_uint64)), ("first", C.POINTER(C.c_uint32)),
                ("second", C.POINTER(C.c_uint32)), ("posterior", C.POINTER(C.c_double))]


EM_KERNELS = 12  # RPVG_HIP_EM_KERNELS


class CEmKernelStats(C.Structure):
    _fields_ = [("ms", C.c_double), ("launches", C.c_uint64), ("problems", C.c_uint64), ("iterations", C.c_uint64),
                ("max_iterations", C.c_uint64), ("alg_bytes", C.c_double)]


class CKernelStats(C.Structure):
    _fields_ = [
        ("em_sparse_ms", C.c_double), ("em_sparse_launches", C.c_uint64), ("em_sparse_alg_bytes", C.c_double),
        ("em_dense_ms", C.c_double), ("em_dense_launches", C.c_uint64), ("em_dense_alg_bytes", C.c_double),
        ("loglik_ms", C.c_double), ("loglik_launches", C.c_uint64), ("loglik_evals", C.c_double),
        ("build_ms", C.c_double), ("build_launches", C.c_uint64),
        ("h2d_ms", C.c_double), ("h2d_bytes", C.c_double),
        ("em_iterations_total", C.c_uint64),
        ("search_pairs_possible", C.c_double), ("search_pairs_table", C.c_double), ("search_pairs_kept", C.c_double),
        ("em_kernel", CEmKernelStats * EM_KERNELS),
        ("collapse_ms", C.c_double), ("busy_ms", C.c_double), ("gibbs_ms", C.c_double),
        ("search_tile_ms", C.c_double), ("search_tile_launches", C.c_uint64),
    ]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "em_kernel"}
        d["em_kernel"] = {em_kernel_name(i): {n: getattr(self.em_kernel[i], n) for n, _ in CEmKernelStats._fields_} for i in range(EM_KERNELS)}
        return d


def em_kernel_name(index: int) -> str:
    f = lib().rpvg_hip_em_kernel_name
    f.restype = C.c_char_p
    return f(index).decode()


_lib = None


def lib() -> C.CDLL:
    """Loads librpvg_hip.so (built by __graft_entry__.build()); raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.rpvg_hip_last_error.restype = C.c_char_p
        for name in EXPORTS:
            getattr(L, name)
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise EngineError(f"{what} failed ({rc}): {lib().rpvg_hip_last_error().decode()}")


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().rpvg_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class CClusterSegment(C.Structure):
    """rpvg_cluster_segment (include/rpvg_batch.h)."""
    _fields_ = [("base", C.c_void_p), ("bytes", C.c_uint64), ("num_rows", C.c_uint32), ("num_groups", C.c_uint32), ("num_entries", C.c_uint32),
                ("num_paths", C.c_uint32), ("num_sources", C.c_uint32), ("has_paths", C.c_uint32), ("total_read_count", C.c_uint64),
                ("row_count_at", C.c_uint64), ("row_noise_at", C.c_uint64), ("row_grp_off_at", C.c_uint64), ("grp_idx_off_at", C.c_uint64),
                ("grp_prob_at", C.c_uint64), ("path_idx_at", C.c_uint64), ("path_group_id_at", C.c_uint64), ("path_source_off_at", C.c_uint64),
                ("source_id_at", C.c_uint64), ("has_columns", C.c_uint32), ("num_columns", C.c_uint32), ("num_column_paths", C.c_uint32),
                ("max_column_paths", C.c_uint32), ("col_count_at", C.c_uint64), ("col_end_at", C.c_uint64), ("col_path_at", C.c_uint64)]


class PinnedSegments:
    """The clusters of a ClusterBatch as one rpvg_cluster_segment each, every segment in a page-locked block of its own
    (rpvg_hip_pinned_alloc) — what the threads of a team calling PathEstimator::estimate() hand to the call combiner."""

    def __init__(self, host: ClusterBatch, with_paths: bool = True, columns=None):
        """columns: per cluster (multiplicities, [path list of every column]) — the caller's own haplotype columns
        (rpvg_cluster_segment::has_columns); the source ids then stay behind."""
        L = lib()
        L.rpvg_hip_pinned_free.argtypes = [C.c_void_p]
        K = host.num_clusters
        self.blocks = []
        self.segments = (CClusterSegment * max(K, 1))()
        self.arrays = []  # numpy views of the blocks, per cluster: name -> array (tests corrupt them)
        for k in range(K):
            r0, r1 = int(host.cluster_row_off[k]), int(host.cluster_row_off[k + 1])
            p0, p1 = int(host.cluster_path_off[k]), int(host.cluster_path_off[k + 1])
            g0, g1 = int(host.row_grp_off[r0]), int(host.row_grp_off[r1])
            e0, e1 = int(host.grp_idx_off[g0]), int(host.grp_idx_off[g1])
            s0, s1 = (int(host.path_source_off[p0]), int(host.path_source_off[p1])) if with_paths else (0, 0)
            pieces = [("row_noise", host.row_noise[r0:r1], np.float64), ("grp_prob", host.grp_prob[g0:g1], np.float64),
                      ("row_count", host.row_count[r0:r1], np.uint32), ("row_grp_off", host.row_grp_off[r0:r1 + 1] - g0, np.uint32),
                      ("grp_idx_off", host.grp_idx_off[g0:g1 + 1] - e0, np.uint32), ("path_idx", host.path_idx[e0:e1], np.uint32)]
            if columns is not None:
                counts, lists = columns[k]
                ends = np.cumsum([len(x) for x in lists]).astype(np.uint32) if lists else np.zeros(0, dtype=np.uint32)
                flat = np.array([p for x in lists for p in x], dtype=np.uint32)
                pieces += [("path_group_id", host.path_group_id[p0:p1], np.uint32), ("col_count", np.array(counts, dtype=np.uint32), np.uint32),
                           ("col_end", ends, np.uint32), ("col_path", flat, np.uint32)]
            elif with_paths:
                pieces += [("path_group_id", host.path_group_id[p0:p1], np.uint32), ("path_source_off", host.path_source_off[p0:p1 + 1] - s0, np.uint32),
                           ("source_id", host.source_id[s0:s1], np.uint32)]
            at, places = 0, {}
            for name, values, dt in pieces:
                places[name] = at
                at += (len(values) * np.dtype(dt).itemsize + 7) & ~7
            block = C.c_void_p()
            _check(L.rpvg_hip_pinned_alloc(C.c_uint64(max(at, 8)), C.byref(block)), "rpvg_hip_pinned_alloc")
            self.blocks.append(block)
            raw = (C.c_ubyte * max(at, 8)).from_address(block.value)
            views = {}
            for name, values, dt in pieces:
                view = np.frombuffer(raw, dtype=dt, count=len(values), offset=places[name])
                view[:] = np.asarray(values).astype(dt)
                views[name] = view
            self.arrays.append(views)
            seg = self.segments[k]
            seg.base, seg.bytes = block.value, max(at, 8)
            seg.num_rows, seg.num_groups, seg.num_entries, seg.num_paths, seg.num_sources = r1 - r0, g1 - g0, e1 - e0, p1 - p0, s1 - s0
            seg.has_paths = 1 if (with_paths and columns is None) else 0
            if columns is not None:
                counts, lists = columns[k]
                seg.has_columns, seg.num_columns, seg.num_column_paths = 1, len(counts), sum(len(x) for x in lists)
                seg.max_column_paths = max([len(x) for x in lists], default=0)
                seg.num_sources = 0
            seg.total_read_count = int(host.row_count[r0:r1].astype(np.uint64).sum())
            for name in places:
                setattr(seg, name + "_at", places[name])
        self.count = K

    def free(self):
        for block in self.blocks:
            lib().rpvg_hip_pinned_free(block)
        self.blocks = []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBatch:
    def __init__(self, ctx: "Context", host: ClusterBatch, compact: bool = False, segments: "PinnedSegments" = None, narrow: bool = False):
        self.ctx = ctx
        self.host = host
        self.handle = C.c_void_p()
        if segments is not None:
            _check(lib().rpvg_hip_batch_upload_segments(ctx.handle, segments.segments, C.c_uint32(segments.count), C.byref(self.handle)),
                   "rpvg_hip_batch_upload_segments")
            return
        cb = host.as_c(compact or narrow, narrow) if narrow else host.as_c(compact)
        _check(lib().rpvg_hip_batch_upload(ctx.handle, C.byref(cb), C.byref(self.handle)), "rpvg_hip_batch_upload")

    def has_source_columns(self) -> bool:
        """Whether the upload formed the haplotype columns of the clusters on the device (path_sources.hip)."""
        return bool(lib().rpvg_hip_batch_has_source_columns(self.handle))

    def cluster_totals(self) -> np.ndarray:
        out = np.zeros(self.host.num_clusters, dtype=np.float64)
        _check(lib().rpvg_hip_batch_cluster_totals(self.handle, C.c_void_p(out.ctypes.data), self.host.num_clusters), "rpvg_hip_batch_cluster_totals")
        return out

    def source_columns(self, cluster: int):
        """(multiplicities, [path list of every column]) of one cluster as the device formed them."""
        ncols, npaths = C.c_uint32(0), C.c_uint32(0)
        _check(lib().rpvg_hip_batch_source_columns_sizes(self.handle, cluster, C.byref(ncols), C.byref(npaths)), "rpvg_hip_batch_source_columns_sizes")
        counts = np.zeros(max(1, ncols.value), dtype=np.uint32)
        ends = np.zeros(max(1, ncols.value), dtype=np.uint32)
        paths = np.zeros(max(1, npaths.value), dtype=np.uint32)
        _check(lib().rpvg_hip_batch_source_columns_get(self.ctx.handle, self.handle, cluster, C.c_void_p(counts.ctypes.data),
                                                       C.c_void_p(ends.ctypes.data), C.c_void_p(paths.ctypes.data)), "rpvg_hip_batch_source_columns_get")
        lists, begin = [], 0
        for c in range(ncols.value):
            lists.append([int(p) for p in paths[begin:ends[c]]])
            begin = int(ends[c])
        return [int(x) for x in counts[:ncols.value]], lists

    def free(self):
        if self.handle:
            lib().rpvg_hip_batch_free(self.ctx.handle, self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceRows:
    """Rows resident on the GPU (rpvg_hip_read_rows)."""

    def __init__(self, ctx: "Context", handle):
        self.ctx = ctx
        self.handle = handle

    def download(self):
        """(ClusterBatch of the rows, build_ms, merge_ms)"""
        from . import rows as rows_mod
        view = CClusterBatch()
        b_ms, m_ms = C.c_double(0), C.c_double(0)
        _check(lib().rpvg_hip_read_rows_view(self.ctx.handle, self.handle, C.byref(view), C.byref(b_ms), C.byref(m_ms)),
               "rpvg_hip_read_rows_view")
        return rows_mod.rows_from_view(view), b_ms.value, m_ms.value

    def to_batch_handle(self) -> C.c_void_p:
        """rpvg_hip_batch made on the device from the rows (caller frees with rpvg_hip_batch_free)."""
        h = C.c_void_p()
        _check(lib().rpvg_hip_read_rows_to_batch(self.ctx.handle, self.handle, C.byref(h)), "rpvg_hip_read_rows_to_batch")
        return h

    def free(self):
        if self.handle:
            lib().rpvg_hip_read_rows_free(self.ctx.handle, self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceAlignments:
    """Alignment-path lists resident on the GPU (rpvg_hip_alignments)."""

    def __init__(self, ctx: "Context", host):
        self.ctx = ctx
        self.host = host
        self.handle = C.c_void_p()
        cb = host.as_c()
        _check(lib().rpvg_hip_alignments_upload(ctx.handle, C.byref(cb), C.byref(self.handle)), "rpvg_hip_alignments_upload")

    def build_rows(self, row_params, merge: bool = True) -> DeviceRows:
        cp = row_params.as_c()
        h = C.c_void_p()
        _check(lib().rpvg_hip_read_rows_build(self.ctx.handle, self.handle, C.byref(cp), C.c_int32(1 if merge else 0), C.byref(h)),
               "rpvg_hip_read_rows_build")
        return DeviceRows(self.ctx, h)

    def free(self):
        if self.handle:
            lib().rpvg_hip_alignments_free(self.ctx.handle, self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def host_register(array: np.ndarray):
    """Page-locks the memory of a numpy array (rpvg_hip_host_register): uploads from it skip the staging copy."""
    if array.nbytes:
        _check(lib().rpvg_hip_host_register(C.c_void_p(array.ctypes.data), C.c_uint64(array.nbytes)), "rpvg_hip_host_register")


def host_unregister(array: np.ndarray):
    if array.nbytes:
        _check(lib().rpvg_hip_host_unregister(C.c_void_p(array.ctypes.data)), "rpvg_hip_host_unregister")


class DeviceGroups:
    def __init__(self, ctx: "Context", batch: DeviceBatch, clusters: Sequence[int], groups: Sequence[Sequence[Sequence[int]]],
                 normalise: bool, collapse_precision: float = 0.0):
        """groups[m] = list of path lists (one per column) for matrix m on clusters[m]; collapse_precision > 0 replays
        readCollapseProbabilityMatrix on the (normalised) matrices."""
        self.ctx = ctx
        self.batch = batch
        cl = np.ascontiguousarray(clusters, dtype=np.uint32)
        if groups is None:  # column c of a matrix = path c of its cluster alone: rpvg_hip_groups_build_single_paths
            self.handle = C.c_void_p()
            _check(lib().rpvg_hip_groups_build_single_paths(ctx.handle, batch.handle, C.c_uint32(len(cl)), C.c_void_p(cl.ctypes.data),
                                                            C.c_int32(1 if normalise else 0), C.c_double(float(collapse_precision)),
                                                            C.byref(self.handle)), "rpvg_hip_groups_build_single_paths")
            return
        goff, gpoff, gp = [0], [0], []
        for cols in groups:
            for paths in cols:
                gp.extend(paths)
                gpoff.append(len(gp))
            goff.append(len(gpoff) - 1)
        goff = np.ascontiguousarray(goff, dtype=np.uint64)
        gpoff = np.ascontiguousarray(gpoff, dtype=np.uint64)
        gp = np.ascontiguousarray(gp, dtype=np.uint32)
        spec = CGroupSpec(len(cl), cl.ctypes.data, goff.ctypes.data, gpoff.ctypes.data, gp.ctypes.data, 1 if normalise else 0,
                          float(collapse_precision))
        self.handle = C.c_void_p()
        _check(lib().rpvg_hip_groups_build(ctx.handle, batch.handle, C.byref(spec), C.byref(self.handle)),
               "rpvg_hip_groups_build")

    def collapse_info(self):
        """(matrices whose runs were replayed, rows that took the values of their run head, matrices sorted as a
        whole, rows that took part in a replay)."""
        out = [C.c_uint32(0) for _ in range(4)]
        _check(lib().rpvg_hip_groups_collapse_info(self.ctx.handle, self.handle, *[C.byref(x) for x in out]),
               "rpvg_hip_groups_collapse_info")
        return tuple(int(x.value) for x in out)

    def loglik(self, matrix, members, divisor: float, add_rowmax=None) -> np.ndarray:
        mt = np.ascontiguousarray(matrix, dtype=np.uint32)
        mem = np.ascontiguousarray(members, dtype=np.uint32)
        if mem.ndim == 1:
            mem = mem.reshape(len(mt), -1)
        width = mem.shape[1]
        out = np.zeros(len(mt), dtype=np.float64)
        flag = None if add_rowmax is None else np.ascontiguousarray(add_rowmax, dtype=np.uint8)
        _check(lib().rpvg_hip_group_loglik(self.ctx.handle, self.handle, C.c_uint32(len(mt)), C.c_void_p(mt.ctypes.data),
                                           C.c_void_p(mem.ctypes.data), C.c_uint32(width), C.c_double(divisor),
                                           C.c_void_p(flag.ctypes.data if flag is not None else None),
                                           C.c_void_p(out.ctypes.data)), "rpvg_hip_group_loglik")
        return out

    def conditionals(self, matrix, others, width: int, divisor: float, num_cols) -> List[np.ndarray]:
        """Per request: log-likelihood of every candidate column given the other width-1 members."""
        mt = np.ascontiguousarray(matrix, dtype=np.uint32)
        oth = np.ascontiguousarray(others, dtype=np.uint32).reshape(len(mt), max(width - 1, 0))
        sizes = [int(num_cols[int(m)]) for m in mt]
        out = np.zeros(sum(sizes), dtype=np.float64)
        _check(lib().rpvg_hip_group_conditionals(self.ctx.handle, self.handle, C.c_uint32(len(mt)), C.c_void_p(mt.ctypes.data),
                                                 C.c_void_p(oth.ctypes.data if oth.size else None), C.c_uint32(width),
                                                 C.c_double(divisor), C.c_void_p(out.ctypes.data)),
               "rpvg_hip_group_conditionals")
        return np.split(out, np.cumsum(sizes)[:-1]) if sizes else []

    def gibbs(self, matrix, group_size: int, num_chains, num_burn_its, num_gibbs_its, log_freq, generator_problems, generator_words):
        """rpvg_hip_group_gibbs: per problem ([sorted member tuples in order of first appearance], counts); plus the words each
        generator gave, the state words of the generators that gave at least 624, and (rounds, conditionals)."""
        mt = np.ascontiguousarray(matrix, dtype=np.uint32)
        ch = np.ascontiguousarray(num_chains, dtype=np.uint32)
        bu = np.ascontiguousarray(num_burn_its, dtype=np.uint32)
        it = np.ascontiguousarray(num_gibbs_its, dtype=np.uint32)
        lf = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64) for x in log_freq]) if len(log_freq) else np.zeros(0))
        goff = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(g) for g in generator_problems])]), dtype=np.uint32)
        gp = np.ascontiguousarray([p for g in generator_problems for p in g], dtype=np.uint32)
        gw = np.ascontiguousarray(generator_words, dtype=np.uint32).reshape(len(generator_problems), 624)
        spec = CGibbsSpec(len(mt), group_size, mt.ctypes.data, ch.ctypes.data, bu.ctypes.data, it.ctypes.data, lf.ctypes.data,
                          len(generator_problems), goff.ctypes.data, gp.ctypes.data, gw.ctypes.data)
        h = C.c_void_p()
        _check(lib().rpvg_hip_group_gibbs(self.ctx.handle, self.handle, C.byref(spec), C.byref(h)), "rpvg_hip_group_gibbs")
        try:
            v = CGibbsSetsView()
            _check(lib().rpvg_hip_gibbs_sets_get(h, C.byref(v)), "rpvg_hip_gibbs_sets_get")
            off = np.ctypeslib.as_array(v.set_off, shape=(len(mt) + 1,)).copy()
            total = int(off[-1])
            first = np.ctypeslib.as_array(v.first, shape=(total,)).copy() if total else np.zeros(0, np.uint32)
            second = np.ctypeslib.as_array(v.second, shape=(total,)).copy() if total else np.zeros(0, np.uint32)
            count = np.ctypeslib.as_array(v.count, shape=(total,)).copy() if total else np.zeros(0, np.uint32)
            if len(generator_problems):
                words = np.ctypeslib.as_array(v.words_consumed, shape=(len(generator_problems),)).copy()
                state = np.ctypeslib.as_array(v.generator_state, shape=(len(generator_problems), 624)).copy()
            else:
                words, state = np.zeros(0, np.uint64), np.zeros((0, 624), np.uint32)
            out = []
            for i in range(len(mt)):
                a, b = int(off[i]), int(off[i + 1])
                sets = [(int(x),) if group_size == 1 else (int(x), int(y)) for x, y in zip(first[a:b], second[a:b])]
                out.append((sets, [int(c) for c in count[a:b]]))
            return out, words, state, (int(v.rounds), int(v.conditionals))
        finally:
            lib().rpvg_hip_gibbs_sets_free(h)

    def bounded_pair_posteriors(self, column_counts, min_rel_likelihood: float):
        """Per matrix: ([(first, second)...], posteriors) of the on-device branch-and-bound."""
        cc = np.ascontiguousarray(column_counts, dtype=np.uint32)
        h = C.c_void_p()
        _check(lib().rpvg_hip_bounded_pair_posteriors(self.ctx.handle, self.handle, C.c_void_p(cc.ctypes.data),
                                                      C.c_double(min_rel_likelihood), C.byref(h)),
               "rpvg_hip_bounded_pair_posteriors")
        try:
            v = CPairPosteriorsView()
            _check(lib().rpvg_hip_pair_posteriors_get(h, C.byref(v)), "rpvg_hip_pair_posteriors_get")
            M = v.num_matrices
            off = np.ctypeslib.as_array(v.pair_off, shape=(M + 1,)).astype(np.int64)
            n = int(off[-1])
            first = np.ctypeslib.as_array(v.first, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
            second = np.ctypeslib.as_array(v.second, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
            post = np.ctypeslib.as_array(v.posterior, shape=(n,)).copy() if n else np.zeros(0)
        finally:
            lib().rpvg_hip_pair_posteriors_free(h)
        return [([(int(a), int(b)) for a, b in zip(first[off[m]:off[m + 1]], second[off[m]:off[m + 1]])],
                 post[off[m]:off[m + 1]]) for m in range(M)]

    def free(self):
        if self.handle:
            lib().rpvg_hip_groups_free(self.ctx.handle, self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream (rpvg_hip_ctx)."""

    def __init__(self, device: int = 0):
        self.handle = C.c_void_p()
        _check(lib().rpvg_hip_create(C.c_int(device), C.byref(self.handle)), "rpvg_hip_create")

    def close(self):
        if self.handle:
            lib().rpvg_hip_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> Tuple[str, int, int]:
        name = C.create_string_buffer(256)
        cus, mem = C.c_uint32(0), C.c_uint64(0)
        _check(lib().rpvg_hip_device_info(self.handle, name, 256, C.byref(cus), C.byref(mem)), "rpvg_hip_device_info")
        return name.value.decode(), cus.value, mem.value

    def synchronize(self):
        _check(lib().rpvg_hip_synchronize(self.handle), "rpvg_hip_synchronize")

    def upload(self, host: ClusterBatch, compact: bool = False, narrow: bool = False) -> DeviceBatch:
        """compact / narrow: the forms of the batch's arrays made for the copy (ClusterBatch.as_c)."""
        return DeviceBatch(self, host, compact, narrow=narrow)

    def upload_segments(self, host: ClusterBatch, segments: PinnedSegments) -> DeviceBatch:
        """The batch from one page-locked segment per cluster (rpvg_hip_batch_upload_segments)."""
        return DeviceBatch(self, host, segments=segments)

    def groups(self, batch: DeviceBatch, clusters, groups, normalise: bool, collapse_precision: float = 0.0) -> DeviceGroups:
        return DeviceGroups(self, batch, clusters, groups, normalise, collapse_precision)

    # ---- EM -----------------------------------------------------------------
    def em_solve(self, batch: DeviceBatch, clusters: Sequence[int], columns: Sequence[Sequence[int]],
                 max_em_its: int = 10000, max_rel_em_conv: float = 1e-3, collapse_precision: float = 0.0):
        """Returns (abundances list per problem, noise_count[P], total_count[P], iterations[P])."""
        P = len(clusters)
        cl = np.ascontiguousarray(clusters, dtype=np.uint32)
        col_off = np.zeros(P + 1, dtype=np.uint64)
        col_off[1:] = np.cumsum([len(c) for c in columns])
        col_path = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint32) for c in columns])
                                        if P else np.zeros(0), dtype=np.uint32)
        abund = np.zeros(int(col_off[-1]), dtype=np.float64)
        noise = np.zeros(P, dtype=np.float64)
        total = np.zeros(P, dtype=np.float64)
        iters = np.zeros(P, dtype=np.uint32)
        probs = CEmProblems(P, cl.ctypes.data, col_off.ctypes.data, col_path.ctypes.data, collapse_precision)
        res = CEmResults(abund.ctypes.data, noise.ctypes.data, total.ctypes.data, iters.ctypes.data)
        _check(lib().rpvg_hip_em_solve(self.handle, batch.handle, C.c_uint32(max_em_its), C.c_double(max_rel_em_conv),
                                       C.byref(probs), C.byref(res)), "rpvg_hip_em_solve")
        off = col_off.astype(np.int64)
        return [abund[off[p]:off[p + 1]] for p in range(P)], noise, total, iters

    def min_path_cover(self, batch: DeviceBatch, clusters: Sequence[int]) -> List[List[int]]:
        cl = np.ascontiguousarray(clusters, dtype=np.uint32)
        n_paths = [int(batch.host.cluster_path_off[k + 1] - batch.host.cluster_path_off[k]) for k in clusters]
        off = np.zeros(len(cl) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(n_paths)
        cover = np.zeros(int(off[-1]), dtype=np.uint32)
        size = np.zeros(len(cl), dtype=np.uint32)
        _check(lib().rpvg_hip_min_path_cover(self.handle, batch.handle, C.c_uint32(len(cl)), C.c_void_p(cl.ctypes.data),
                                             C.c_void_p(off.ctypes.data), C.c_void_p(cover.ctypes.data),
                                             C.c_void_p(size.ctypes.data)), "rpvg_hip_min_path_cover")
        return [[int(x) for x in cover[int(off[i]):int(off[i]) + int(size[i])]] for i in range(len(cl))]

    # ---- dense ----------------------------------------------------------------
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _check(lib().rpvg_hip_malloc(self.handle, C.c_uint64(nbytes), C.byref(p)), "rpvg_hip_malloc")
        return p.value

    def free(self, ptr: int):
        _check(lib().rpvg_hip_free(self.handle, C.c_void_p(ptr)), "rpvg_hip_free")

    def h2d(self, dst: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        _check(lib().rpvg_hip_memcpy_h2d(self.handle, C.c_void_p(dst), C.c_void_p(a.ctypes.data), C.c_uint64(a.nbytes)),
               "rpvg_hip_memcpy_h2d")

    def d2h(self, src: int, shape, dtype=np.float64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        _check(lib().rpvg_hip_memcpy_d2h(self.handle, C.c_void_p(out.ctypes.data), C.c_void_p(src), C.c_uint64(out.nbytes)),
               "rpvg_hip_memcpy_d2h")
        return out

    def em_dense(self, d_matrix: int, R: int, Cn: int, ld: int, d_counts: int, total: float, max_em_its: int = 10000,
                 max_rel_em_conv: float = 1e-3, sharded: bool = False):
        """EM on a resident dense matrix.  sharded=True: the R rows are this rank's share of one cluster spread
        over the ranks of the context's communicator (comm_init) and `total` is the whole cluster's read count."""
        ab = np.zeros(Cn - 1, dtype=np.float64)
        noise = C.c_double(0)
        its = C.c_uint32(0)
        fn, name = ((lib().rpvg_hip_em_dense_sharded, "rpvg_hip_em_dense_sharded") if sharded
                    else (lib().rpvg_hip_em_dense, "rpvg_hip_em_dense"))
        _check(fn(self.handle, C.c_void_p(d_matrix), C.c_uint64(R), C.c_uint32(Cn), C.c_uint64(ld),
                  C.c_void_p(d_counts), C.c_double(total), C.c_uint32(max_em_its),
                  C.c_double(max_rel_em_conv), C.c_void_p(ab.ctypes.data), C.byref(noise),
                  C.byref(its)), name)
        return ab, noise.value, its.value

    # ---- communicator (RCCL) ----------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(lib().rpvg_hip_comm_unique_id(buf), "rpvg_hip_comm_unique_id")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, world_size: int, rank: int):
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().rpvg_hip_comm_init(self.handle, buf, C.c_int(world_size), C.c_int(rank)), "rpvg_hip_comm_init")

    def comm_destroy(self):
        _check(lib().rpvg_hip_comm_destroy(self.handle), "rpvg_hip_comm_destroy")

    def comm_init_all(self):
        """A communicator over the contexts of this process (rpvg_hip_comm_init_all) — here: this one context."""
        arr = (C.c_void_p * 1)(self.handle)
        _check(lib().rpvg_hip_comm_init_all(arr, C.c_int(1)), "rpvg_hip_comm_init_all")

    def gather(self, local: np.ndarray, counts) -> np.ndarray:
        """rpvg_hip_gather: this rank's values in, the values of all ranks (counts[r] each) out."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        cnt = np.ascontiguousarray(counts, dtype=np.uint64)
        out = np.zeros(int(cnt.sum()), dtype=np.float64)
        _check(lib().rpvg_hip_gather(self.handle, C.c_void_p(local.ctypes.data), C.c_uint64(local.size), C.c_void_p(cnt.ctypes.data),
                                     C.c_void_p(out.ctypes.data)), "rpvg_hip_gather")
        return out

    def allreduce_sum_f64(self, d_buf: int, n: int):
        _check(lib().rpvg_hip_comm_allreduce_sum_f64(self.handle, C.c_void_p(d_buf), C.c_uint64(n)),
               "rpvg_hip_comm_allreduce_sum_f64")

    def dense_from_cluster(self, batch: DeviceBatch, cluster: int, d_matrix: int, ld: int, d_counts: int) -> float:
        total = C.c_double(0)
        _check(lib().rpvg_hip_dense_from_cluster(self.handle, batch.handle, C.c_uint32(cluster), C.c_void_p(d_matrix),
                                                 C.c_uint64(ld), C.c_void_p(d_counts), C.byref(total)),
               "rpvg_hip_dense_from_cluster")
        return total.value

    def synth_dense_cluster(self, seed: int, R: int, N: int, d_matrix: int, ld: int, d_counts: int):
        _check(lib().rpvg_hip_synth_dense_cluster(self.handle, C.c_uint64(seed), C.c_uint64(R), C.c_uint32(N),
                                                  C.c_void_p(d_matrix), C.c_uint64(ld), C.c_void_p(d_counts)),
               "rpvg_hip_synth_dense_cluster")

    def synth_dense_rows(self, seed: int, row_begin: int, R: int, N: int, d_matrix: int, ld: int, d_counts: int):
        """Rows [row_begin, row_begin + R) of the synthetic dense cluster `seed` (a rank's shard)."""
        _check(lib().rpvg_hip_synth_dense_rows(self.handle, C.c_uint64(seed), C.c_uint64(row_begin), C.c_uint64(R),
                                               C.c_uint32(N), C.c_void_p(d_matrix), C.c_uint64(ld), C.c_void_p(d_counts)),
               "rpvg_hip_synth_dense_rows")

    # ---- row construction (include/rpvg_rows.h) ------------------------------------
    def upload_alignments(self, align_batch) -> "DeviceAlignments":
        return DeviceAlignments(self, align_batch)

    def build_rows(self, align_batch, row_params, merge: bool = True):
        """addPathProbs for every read (+ sort / merge) on the GPU -> (ClusterBatch of rows, build_ms, merge_ms).
        align_batch: an AlignmentBatch (uploaded for this call) or DeviceAlignments (already resident)."""
        dev = align_batch if isinstance(align_batch, DeviceAlignments) else DeviceAlignments(self, align_batch)
        try:
            rows = dev.build_rows(row_params, merge)
            try:
                return rows.download()
            finally:
                rows.free()
        finally:
            if dev is not align_batch:
                dev.free()

    # ---- path clustering ----------------------------------------------------------
    def path_clusters(self, num_paths: int, sets):
        """sets: id sets (lists of path ids) -> (path_to_cluster[num_paths], [members of cluster 0, 1, ...])."""
        off = np.zeros(len(sets) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in sets])
        flat = np.ascontiguousarray([p for x in sets for p in x], dtype=np.uint32)
        return self.path_clusters_flat(num_paths, off, flat)

    def path_clusters_flat(self, num_paths: int, set_off: np.ndarray, set_path: np.ndarray):
        set_off = np.ascontiguousarray(set_off, dtype=np.uint64)
        set_path = np.ascontiguousarray(set_path, dtype=np.uint32)
        p2c = np.zeros(max(num_paths, 1), dtype=np.uint32)
        coff = np.zeros(num_paths + 1, dtype=np.uint64)
        cpaths = np.zeros(max(num_paths, 1), dtype=np.uint32)
        nc = C.c_uint32(0)
        _check(lib().rpvg_hip_path_clusters(self.handle, C.c_uint32(num_paths), C.c_uint64(len(set_off) - 1),
                                            C.c_void_p(set_off.ctypes.data), C.c_void_p(set_path.ctypes.data if set_path.size else None),
                                            C.c_void_p(p2c.ctypes.data), C.byref(nc), C.c_void_p(coff.ctypes.data),
                                            C.c_void_p(cpaths.ctypes.data)), "rpvg_hip_path_clusters")
        k = nc.value
        members = [cpaths[int(coff[c]):int(coff[c + 1])].tolist() for c in range(k)]
        return p2c[:num_paths].copy(), members

    def debug_log(self, x: np.ndarray, use_table: bool = True) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros_like(x)
        _check(lib().rpvg_hip_debug_log(self.handle, C.c_uint64(x.size), C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data),
                                        C.c_int32(1 if use_table else 0)), "rpvg_hip_debug_log")
        return out

    # ---- stats ----------------------------------------------------------------
    def stats(self) -> dict:
        s = CKernelStats()
        _check(lib().rpvg_hip_stats_get(self.handle, C.byref(s)), "rpvg_hip_stats_get")
        return s.as_dict()

    def reset_stats(self):
        _check(lib().rpvg_hip_stats_reset(self.handle), "rpvg_hip_stats_reset")
