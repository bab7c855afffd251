"""ctypes loader for the CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package ``rpvg_amd`` never does.

"parity unpinned" — see oracle/rpvg_oracle.hpp for what that covers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

from rpvg_amd.batch import CClusterBatch, CEstimatesView, CParams, ClusterBatch, ClusterEstimates, decode_view

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librpvg_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/_build/librpvg_oracle.so with the committed Makefile (g++ only)."""
    srcs = [os.path.join(_HERE, f) for f in ("rpvg_oracle.cpp", "rpvg_oracle_capi.cpp", "rpvg_oracle.hpp")]
    srcs.append(os.path.join(_HERE, "..", "include", "rpvg_batch.h"))
    stale = force or not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.rpvg_oracle_run.restype = C.c_void_p
        L.rpvg_oracle_run.argtypes = [C.c_char_p, C.POINTER(CParams), C.POINTER(CClusterBatch), C.c_int,
                                      C.POINTER(C.c_double)]
        L.rpvg_oracle_view.argtypes = [C.c_void_p, C.POINTER(CEstimatesView)]
        L.rpvg_oracle_free.argtypes = [C.c_void_p]
        L.rpvg_oracle_em_dense.restype = C.c_uint32
        L.rpvg_oracle_em_dense.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_double,
                                           C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double)]
        L.rpvg_oracle_min_path_cover.restype = C.c_uint32
        L.rpvg_oracle_min_path_cover.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rpvg_oracle_group_posteriors.restype = C.c_uint32
        L.rpvg_oracle_group_posteriors.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_uint32, C.c_int, C.c_double, C.c_uint32,
                                                   C.c_void_p, C.c_void_p]
        L.rpvg_oracle_num_permutations.restype = C.c_uint32
        L.rpvg_oracle_num_permutations.argtypes = [C.c_void_p, C.c_uint32]
        L.rpvg_oracle_add_log.restype = C.c_double
        L.rpvg_oracle_add_log.argtypes = [C.c_double, C.c_double]
        L.rpvg_oracle_double_compare.restype = C.c_int
        L.rpvg_oracle_double_compare.argtypes = [C.c_double, C.c_double]
        L.rpvg_oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def run(model: str, params: CParams, batch: ClusterBatch, num_threads: int = 1) -> Tuple[List[ClusterEstimates], float]:
    """estimate() over every cluster of the batch (OpenMP over clusters as src/main.cpp:829)."""
    L = lib()
    cb = batch.as_c()
    secs = C.c_double(0)
    h = L.rpvg_oracle_run(model.encode(), C.byref(params), C.byref(cb), int(num_threads), C.byref(secs))
    try:
        view = CEstimatesView()
        L.rpvg_oracle_view(h, C.byref(view))
        out = decode_view(view)
    finally:
        L.rpvg_oracle_free(h)
    return out, secs.value


class RawRun:
    """estimate() over a batch, keeping the C view alive (to hand the oracle's estimates to a writer)."""

    def __init__(self, model: str, params: CParams, batch: ClusterBatch, num_threads: int = 1):
        L = lib()
        self._batch = batch
        cb = batch.as_c()
        self._h = L.rpvg_oracle_run(model.encode(), C.byref(params), C.byref(cb), int(num_threads), None)
        self.view = CEstimatesView()
        L.rpvg_oracle_view(self._h, C.byref(self.view))
        self.estimates = decode_view(self.view)

    def close(self):
        if self._h:
            lib().rpvg_oracle_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def em_dense(P: np.ndarray, counts: np.ndarray, max_em_its: int = 10000, max_rel_em_conv: float = 1e-3):
    """EMAbundanceEstimator on a normalised dense R x C matrix (last column = noise).

    Returns (abundances[C-1], noise_count, total_count, iterations, seconds)."""
    L = lib()
    Pf = np.asfortranarray(P, dtype=np.float64)
    R, Cn = Pf.shape
    c = np.ascontiguousarray(counts, dtype=np.float64)
    ab = np.zeros(Cn - 1, dtype=np.float64)
    noise, total, secs = C.c_double(0), C.c_double(0), C.c_double(0)
    its = L.rpvg_oracle_em_dense(Pf.ctypes.data, R, Cn, c.ctypes.data, max_em_its, max_rel_em_conv, ab.ctypes.data,
                                 C.byref(noise), C.byref(total), C.byref(secs))
    return ab, noise.value, total.value, int(its), secs.value


def min_path_cover(cover: np.ndarray, read_counts, path_weights) -> List[int]:
    L = lib()
    cv = np.ascontiguousarray(cover, dtype=np.uint8)
    R, N = cv.shape
    rc = np.ascontiguousarray(read_counts, dtype=np.float64)
    pw = np.ascontiguousarray(path_weights, dtype=np.float64)
    out = np.zeros(N, dtype=np.uint32)
    n = L.rpvg_oracle_min_path_cover(cv.ctypes.data, R, N, rc.ctypes.data, pw.ctypes.data, out.ctypes.data)
    return [int(x) for x in out[:n]]


def group_posteriors(P: np.ndarray, noise, counts, path_counts, group_size: int, bounded: bool = False,
                     min_rel_lik: float = 1e-8):
    """Full / Bounded group posteriors.  Returns ([members tuple...], posteriors)."""
    L = lib()
    Pf = np.asfortranarray(P, dtype=np.float64)
    R, N = Pf.shape
    nz = np.ascontiguousarray(noise, dtype=np.float64)
    c = np.ascontiguousarray(counts, dtype=np.float64)
    pc = np.ascontiguousarray(path_counts, dtype=np.uint32)
    from math import comb
    max_sets = comb(N + group_size - 1, group_size)
    mem = np.zeros(max_sets * group_size, dtype=np.uint32)
    post = np.zeros(max_sets, dtype=np.float64)
    n = L.rpvg_oracle_group_posteriors(Pf.ctypes.data, R, N, nz.ctypes.data, c.ctypes.data, pc.ctypes.data, group_size,
                                       1 if bounded else 0, min_rel_lik, max_sets, mem.ctypes.data, post.ctypes.data)
    assert n != 0xFFFFFFFF
    sets = [tuple(int(x) for x in mem[i * group_size:(i + 1) * group_size]) for i in range(n)]
    return sets, post[:n].copy()


def num_permutations(values) -> int:
    v = np.ascontiguousarray(values, dtype=np.uint32)
    return int(lib().rpvg_oracle_num_permutations(v.ctypes.data, len(v)))


def add_log(x: float, y: float) -> float:
    return float(lib().rpvg_oracle_add_log(x, y))


def max_threads() -> int:
    return int(lib().rpvg_oracle_max_threads())


# ---- row construction -----------------------------------------------------------

def frag_length_table(loc: float, scale: float, shape: float = 0.0, sd_max_multi: int = 10) -> np.ndarray:
    """FragmentLengthDist::logProb(v) for v = 0..65535 (src/fragment_length_dist.cpp:385-427)."""
    L = lib()
    L.rpvg_oracle_frag_length_table.argtypes = [C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_void_p]
    out = np.zeros(65536, dtype=np.float64)
    L.rpvg_oracle_frag_length_table(loc, scale, shape, sd_max_multi, out.ctypes.data)
    return out


def build_rows(align_batch, row_params, merge: bool = True, num_threads: int = 1):
    """addPathProbs for every read of every cluster (+ sort / quickMergeIdentical when merge) -> (ClusterBatch, seconds)."""
    from rpvg_amd import rows as rows_mod
    L = lib()
    L.rpvg_oracle_build_rows.restype = C.c_void_p
    L.rpvg_oracle_build_rows.argtypes = [C.POINTER(rows_mod.CAlignmentBatch), C.POINTER(rows_mod.CRowParams), C.c_int, C.c_int,
                                         C.POINTER(C.c_double)]
    L.rpvg_oracle_rows_view.argtypes = [C.c_void_p, C.POINTER(CClusterBatch)]
    L.rpvg_oracle_rows_free.argtypes = [C.c_void_p]
    cb, cp = align_batch.as_c(), row_params.as_c()
    secs = C.c_double(0)
    h = L.rpvg_oracle_build_rows(C.byref(cb), C.byref(cp), 1 if merge else 0, int(num_threads), C.byref(secs))
    try:
        view = CClusterBatch()
        L.rpvg_oracle_rows_view(h, C.byref(view))
        return rows_mod.rows_from_view(view), secs.value
    finally:
        L.rpvg_oracle_rows_free(h)


def path_clusters(num_paths: int, sets):
    """PathClusters over id sets -> (path_to_cluster, [members...], seconds)."""
    L = lib()
    L.rpvg_oracle_path_clusters.restype = C.c_uint32
    L.rpvg_oracle_path_clusters.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_double)]
    off = np.zeros(len(sets) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in sets])
    flat = np.ascontiguousarray([p for x in sets for p in x], dtype=np.uint32)
    return path_clusters_flat(num_paths, off, flat)


def path_clusters_flat(num_paths: int, set_off, set_path):
    L = lib()
    L.rpvg_oracle_path_clusters.restype = C.c_uint32
    L.rpvg_oracle_path_clusters.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_double)]
    set_off = np.ascontiguousarray(set_off, dtype=np.uint64)
    set_path = np.ascontiguousarray(set_path, dtype=np.uint32)
    p2c = np.zeros(max(num_paths, 1), dtype=np.uint32)
    coff = np.zeros(num_paths + 1, dtype=np.uint64)
    cpaths = np.zeros(max(num_paths, 1), dtype=np.uint32)
    secs = C.c_double(0)
    k = L.rpvg_oracle_path_clusters(num_paths, len(set_off) - 1, set_off.ctypes.data, set_path.ctypes.data if set_path.size else None,
                                    p2c.ctypes.data, coff.ctypes.data, cpaths.ctypes.data, C.byref(secs))
    members = [cpaths[int(coff[c]):int(coff[c + 1])].tolist() for c in range(k)]
    return p2c[:num_paths].copy(), members, secs.value
