"""Synthetic pantranscriptome batches (bench / tests): ctypes view of the C++ generator in
``rpvg_amd/host/synth_pantranscriptome.cpp`` (the model is described there)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine as _engine
from .batch import CClusterBatch, ClusterBatch


class CSynthConfig(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("num_clusters", C.c_uint32), ("total_paths", C.c_uint64), ("total_reads", C.c_uint64),
        ("num_haplotypes", C.c_uint32), ("max_cluster_paths", C.c_uint32), ("cluster_paths_sigma", C.c_double),
        ("read_mass_sigma", C.c_double), ("tie_prob", C.c_double), ("pathless_read_frac", C.c_double),
        ("keep_alignments", C.c_uint32),
    ]


# BASELINE.json configs[2]: 10M read pairs x 200k paths in ~5k clusters
FULL = dict(seed=3, num_clusters=5000, total_paths=200000, total_reads=10000000)


def _bind():
    L = _engine.lib()
    L.rpvg_amd_synth_default_config.restype = CSynthConfig
    L.rpvg_amd_synth_generate.restype = C.c_void_p
    L.rpvg_amd_synth_generate.argtypes = [C.POINTER(CSynthConfig)]
    L.rpvg_amd_synth_view.argtypes = [C.c_void_p, C.POINTER(CClusterBatch)]
    L.rpvg_amd_synth_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 5
    L.rpvg_amd_synth_free.argtypes = [C.c_void_p]
    L.rpvg_amd_rows_from_likelihoods.restype = C.c_void_p
    L.rpvg_amd_rows_from_likelihoods.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_double]
    return L


def _to_batch(L, h) -> ClusterBatch:
    view = CClusterBatch()
    L.rpvg_amd_synth_view(h, C.byref(view))
    sizes = [C.c_uint64(0) for _ in range(5)]
    L.rpvg_amd_synth_sizes(h, *[C.byref(s) for s in sizes])
    R, G, NNZ, P, S = (int(s.value) for s in sizes)
    K = view.num_clusters

    def arr(ptr, n, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

    return ClusterBatch(
        arr(view.cluster_row_off, K + 1, np.uint64), arr(view.cluster_path_off, K + 1, np.uint64),
        arr(view.row_count, R, np.uint32), arr(view.row_noise, R, np.float64), arr(view.row_grp_off, R + 1, np.uint64),
        arr(view.grp_prob, G, np.float64), arr(view.grp_idx_off, G + 1, np.uint64), arr(view.path_idx, NNZ, np.uint32),
        arr(view.path_group_id, P, np.uint32), arr(view.path_source_count, P, np.uint32),
        arr(view.path_source_off, P + 1, np.uint64), arr(view.source_id, S, np.uint32),
        arr(view.path_effective_length, P, np.float64))


def rows_from_likelihoods(num_paths: int, reads, prob_precision: float = 1e-8) -> ClusterBatch:
    """reads: [(count, noise, {path: likelihood})...] -> one cluster of finished, sorted, merged rows
    (C++ ReadPathProbabilities::fromPathLikelihoods + sortAndMergeReadPathProbabilities)."""
    L = _bind()
    cnt = np.ascontiguousarray([r[0] for r in reads], dtype=np.uint32)
    noise = np.ascontiguousarray([r[1] for r in reads], dtype=np.float64)
    off, paths, vals = [0], [], []
    for r in reads:
        for p in sorted(r[2]):
            paths.append(p)
            vals.append(r[2][p])
        off.append(len(paths))
    off = np.ascontiguousarray(off, dtype=np.uint64)
    paths = np.ascontiguousarray(paths, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    h = L.rpvg_amd_rows_from_likelihoods(num_paths, len(reads), cnt.ctypes.data, noise.ctypes.data, off.ctypes.data,
                                         paths.ctypes.data, vals.ctypes.data, prob_precision)
    try:
        return _to_batch(L, h)
    finally:
        L.rpvg_amd_synth_free(h)


def generate(seed: int = 3, num_clusters: int = 5000, total_paths: int = 200000, total_reads: int = 10000000,
             **overrides) -> ClusterBatch:
    L = _bind()
    cfg = L.rpvg_amd_synth_default_config()
    cfg.seed, cfg.num_clusters, cfg.total_paths, cfg.total_reads = seed, num_clusters, total_paths, total_reads
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise KeyError(k)
        setattr(cfg, k, v)
    h = L.rpvg_amd_synth_generate(C.byref(cfg))
    try:
        return _to_batch(L, h)
    finally:
        L.rpvg_amd_synth_free(h)


# what the generator's reads look like as alignments: best score, alignment length, fragment length
SYNTH_FRAG_LENGTH = 300


def generate_with_alignments(seed: int = 3, num_clusters: int = 5000, total_paths: int = 200000, total_reads: int = 10000000,
                             **overrides):
    """(ClusterBatch of finished rows, AlignmentBatch of the same reads as alignment-path lists): the second is what
    row construction (include/rpvg_rows.h) starts from; the first is the generator's own result for them."""
    from . import rows as rows_mod
    L = _bind()
    L.rpvg_amd_synth_alignments_view.argtypes = [C.c_void_p, C.POINTER(rows_mod.CAlignmentBatch)]
    cfg = L.rpvg_amd_synth_default_config()
    cfg.seed, cfg.num_clusters, cfg.total_paths, cfg.total_reads = seed, num_clusters, total_paths, total_reads
    cfg.keep_alignments = 1
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise KeyError(k)
        setattr(cfg, k, v)
    h = L.rpvg_amd_synth_generate(C.byref(cfg))
    try:
        batch = _to_batch(L, h)
        view = rows_mod.CAlignmentBatch()
        assert L.rpvg_amd_synth_alignments_view(h, C.byref(view)) == 0
        K = view.num_clusters

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

        cro = arr(view.cluster_read_off, K + 1, np.uint64)
        N = int(cro[-1])
        rao = arr(view.read_align_off, N + 1, np.uint64)
        A = int(rao[-1])
        apo = arr(view.align_path_off, A + 1, np.uint64)
        E = int(apo[-1])
        P = batch.num_paths
        aligns = rows_mod.AlignmentBatch(
            cro, batch.cluster_path_off.copy(), batch.path_effective_length.copy(), batch.path_source_count.copy(), None, None,
            arr(view.read_count, N, np.uint32), arr(view.read_min_mapq, N, np.uint8), arr(view.read_noise_score, N, np.int32), rao,
            arr(view.align_score_sum, A, np.int32), arr(view.align_length, A, np.uint16), arr(view.align_frag_length, A, np.uint16),
            apo, arr(view.align_path_idx, E, np.uint32))
        assert len(aligns.path_effective_length) == P
        return batch, aligns
    finally:
        L.rpvg_amd_synth_free(h)
