"""Flat alignment-path batches (include/rpvg_rows.h): the input of row construction, the step right before the
inference hot path (ReadPathProbabilities::addPathProbs + sort/merge, src/read_path_probabilities.cpp:39-322,
src/main.cpp:889-973).  Marshalling for tests and bench only."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from .batch import CClusterBatch, ClusterBatch, f64p, u32p, u64p, _ptr

FRAG_LENGTH_TABLE_SIZE = 65536
u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
i32p = C.POINTER(C.c_int32)
INT32_LOWEST = -2147483648


class CAlignmentBatch(C.Structure):
    _fields_ = [
        ("num_clusters", C.c_uint32), ("cluster_read_off", u64p), ("cluster_path_off", u64p),
        ("path_effective_length", f64p), ("path_source_count", u32p), ("path_group", u32p), ("cluster_group_off", u64p),
        ("read_count", u32p), ("read_min_mapq", u8p), ("read_noise_score", i32p), ("read_align_off", u64p),
        ("align_score_sum", i32p), ("align_length", u16p), ("align_frag_length", u16p), ("align_path_off", u64p),
        ("align_path_idx", u32p),
    ]


class CRowParams(C.Structure):
    _fields_ = [("prob_precision", C.c_double), ("min_noise_prob", C.c_double), ("is_single_end", C.c_int32),
                ("frag_length_log_prob", f64p)]


@dataclass
class RowParams:
    prob_precision: float = 1e-8
    min_noise_prob: float = 1e-4
    is_single_end: bool = False
    frag_length_log_prob: Optional[np.ndarray] = None  # f64 [65536]

    def as_c(self) -> CRowParams:
        if not self.is_single_end:
            assert self.frag_length_log_prob is not None and len(self.frag_length_log_prob) == FRAG_LENGTH_TABLE_SIZE
            self.frag_length_log_prob = np.ascontiguousarray(self.frag_length_log_prob, dtype=np.float64)
        ptr = _ptr(self.frag_length_log_prob, f64p) if self.frag_length_log_prob is not None else None
        return CRowParams(self.prob_precision, self.min_noise_prob, 1 if self.is_single_end else 0, ptr)


@dataclass
class AlignmentBatch:
    cluster_read_off: np.ndarray
    cluster_path_off: np.ndarray
    path_effective_length: np.ndarray
    path_source_count: np.ndarray
    path_group: Optional[np.ndarray]
    cluster_group_off: Optional[np.ndarray]
    read_count: np.ndarray
    read_min_mapq: np.ndarray
    read_noise_score: np.ndarray
    read_align_off: np.ndarray
    align_score_sum: np.ndarray
    align_length: np.ndarray
    align_frag_length: np.ndarray
    align_path_off: np.ndarray
    align_path_idx: np.ndarray

    _DTYPES = dict(cluster_read_off=np.uint64, cluster_path_off=np.uint64, path_effective_length=np.float64,
                   path_source_count=np.uint32, path_group=np.uint32, cluster_group_off=np.uint64, read_count=np.uint32,
                   read_min_mapq=np.uint8, read_noise_score=np.int32, read_align_off=np.uint64,
                   align_score_sum=np.int32, align_length=np.uint16, align_frag_length=np.uint16,
                   align_path_off=np.uint64, align_path_idx=np.uint32)

    def __post_init__(self):
        for name, dt in self._DTYPES.items():
            v = getattr(self, name)
            if v is not None:
                setattr(self, name, np.ascontiguousarray(v, dtype=dt))

    @property
    def num_clusters(self) -> int:
        return len(self.cluster_read_off) - 1

    @property
    def num_reads(self) -> int:
        return int(self.cluster_read_off[-1])

    @property
    def total_reads(self) -> int:
        return int(self.read_count.astype(np.uint64).sum())

    def as_c(self) -> CAlignmentBatch:
        def opt(a, ty):
            return _ptr(a, ty) if a is not None else None
        return CAlignmentBatch(
            self.num_clusters, _ptr(self.cluster_read_off, u64p), _ptr(self.cluster_path_off, u64p),
            _ptr(self.path_effective_length, f64p), _ptr(self.path_source_count, u32p), opt(self.path_group, u32p),
            opt(self.cluster_group_off, u64p), _ptr(self.read_count, u32p), _ptr(self.read_min_mapq, u8p),
            _ptr(self.read_noise_score, i32p), _ptr(self.read_align_off, u64p), _ptr(self.align_score_sum, i32p),
            _ptr(self.align_length, u16p), _ptr(self.align_frag_length, u16p), _ptr(self.align_path_off, u64p),
            _ptr(self.align_path_idx, u32p))

    @staticmethod
    def from_clusters(clusters: Sequence[dict]) -> "AlignmentBatch":
        """clusters: [{"paths": [{"effective_length", "source_count"?, "group"?}...],
                       "reads": [{"count", "min_mapq", "noise_score", "aligns": [(score_sum, align_length, frag_length, [path idx...])...]}...]}...]
        A "group" key on the paths turns on collapsing (every path of the batch needs one)."""
        cro, cpo, cgo = [0], [0], [0]
        pel, psc, pgr = [], [], []
        rc, rm, rn, rao = [], [], [], [0]
        asc, al, afl, apo, api = [], [], [], [0], []
        collapse = any("group" in p for cl in clusters for p in cl["paths"])
        for cl in clusters:
            for p in cl["paths"]:
                pel.append(p["effective_length"])
                psc.append(p.get("source_count", 1))
                if collapse:
                    pgr.append(p["group"])
            cpo.append(len(pel))
            if collapse:
                cgo.append(cgo[-1] + (max(p["group"] for p in cl["paths"]) + 1 if cl["paths"] else 0))
            for rd in cl["reads"]:
                rc.append(rd["count"])
                rm.append(rd["min_mapq"])
                rn.append(rd["noise_score"])
                for (score, alen, flen, idxs) in rd["aligns"]:
                    asc.append(score)
                    al.append(alen)
                    afl.append(flen)
                    api.extend(sorted(idxs))
                    apo.append(len(api))
                rao.append(len(asc))
            cro.append(len(rc))
        return AlignmentBatch(cro, cpo, pel, psc, pgr if collapse else None, cgo if collapse else None, rc, rm, rn, rao,
                              asc, al, afl, apo, api)


def rows_from_view(view: CClusterBatch) -> ClusterBatch:
    """Copies the row arrays of a rpvg_cluster_batch view (path metadata is not part of it)."""
    K = view.num_clusters

    def arr(ptr, n, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

    cro = arr(view.cluster_row_off, K + 1, np.uint64)
    cpo = arr(view.cluster_path_off, K + 1, np.uint64)
    R = int(cro[-1])
    rgo = arr(view.row_grp_off, R + 1, np.uint64)
    G = int(rgo[-1])
    gio = arr(view.grp_idx_off, G + 1, np.uint64)
    P = int(cpo[-1])
    return ClusterBatch(cro, cpo, arr(view.row_count, R, np.uint32), arr(view.row_noise, R, np.float64), rgo,
                        arr(view.grp_prob, G, np.float64), gio, arr(view.path_idx, int(gio[-1]), np.uint32),
                        np.zeros(P, np.uint32), np.ones(P, np.uint32), np.zeros(P + 1, np.uint64), np.zeros(0, np.uint32),
                        np.zeros(P, np.float64))
