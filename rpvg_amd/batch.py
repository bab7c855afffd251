"""Flat (C-ABI) representation of a batch of path clusters.

Python-side mirror of ``include/rpvg_batch.h``: numpy arrays for the ragged
batch, ctypes structs that point into them, and a decoder for the estimates
view.  This is harness plumbing for tests/bench (the host engine itself is
C++, ``rpvg_amd/host``); nothing here computes anything.

Reference types being flattened: ``ReadPathProbabilities``
(src/read_path_probabilities.hpp:39-43), ``PathInfo`` and
``PathClusterEstimates`` (src/path_cluster_estimates.hpp:15-57).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)


class CClusterBatch(C.Structure):
    _fields_ = [
        ("num_clusters", C.c_uint32),
        ("cluster_row_off", u64p),
        ("cluster_path_off", u64p),
        ("row_count", u32p),
        ("row_noise", f64p),
        ("row_grp_off", u64p),
        ("grp_prob", f64p),
        ("grp_idx_off", u64p),
        ("path_idx", u32p),
        ("path_group_id", u32p),
        ("path_source_count", u32p),
        ("path_source_off", u64p),
        ("source_id", u32p),
        ("path_effective_length", f64p),
        ("row_grp_off32", u32p),
        ("grp_idx_off32", u32p),
        ("row_grp_count8", u8p),
        ("grp_idx_count8", u8p),
        ("num_groups", C.c_uint64),
        ("num_entries", C.c_uint64),
        ("path_idx16", u16p),
        ("source_id16", u16p),
        ("row_count8", u8p),
        ("row_count_escape_row", u32p),
        ("row_count_escape_count", u32p),
        ("num_row_count_escapes", C.c_uint64),
        ("row_noise16", u16p),
        ("row_noise_table", f64p),
        ("num_row_noise_values", C.c_uint64),
    ]


class CParams(C.Structure):
    _fields_ = [
        ("max_em_its", C.c_uint32),
        ("max_rel_em_conv", C.c_double),
        ("num_gibbs_samples", C.c_uint32),
        ("gibbs_thin_its", C.c_uint32),
        ("prob_precision", C.c_double),
        ("ploidy", C.c_uint32),
        ("min_hap_prob", C.c_double),
        ("ind_hap_inference", C.c_int32),
        ("use_hap_gibbs", C.c_int32),
        ("rng_seed", C.c_uint32),
    ]


class CEstimatesView(C.Structure):
    _fields_ = [
        ("num_clusters", C.c_uint32),
        ("set_off", u64p),
        ("member_off", u64p),
        ("members", u32p),
        ("posteriors", f64p),
        ("abund_off", u64p),
        ("abundances", f64p),
        ("noise_count", f64p),
        ("total_count", f64p),
        ("em_off", u64p),
        ("em_iters", u32p),
        ("em_col_off", u64p),
        ("em_cols", u32p),
        ("gibbs_off", u64p),
        ("gibbs_path_off", u64p),
        ("gibbs_path", u32p),
        ("gibbs_noise_off", u64p),
        ("gibbs_noise", f64p),
        ("gibbs_abund_off", u64p),
        ("gibbs_abund", f64p),
    ]


def make_params(**kw) -> CParams:
    """Defaults of src/main.cpp:402-418."""
    p = CParams(10000, 0.001, 0, 25, 1e-8, 2, 0.001, 0, 0, 0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


@dataclass
class ClusterBatch:
    """K clusters back to back (all arrays C-contiguous, exact dtypes)."""

    cluster_row_off: np.ndarray  # u64 [K+1]
    cluster_path_off: np.ndarray  # u64 [K+1]
    row_count: np.ndarray  # u32 [R]
    row_noise: np.ndarray  # f64 [R]
    row_grp_off: np.ndarray  # u64 [R+1]
    grp_prob: np.ndarray  # f64 [G]
    grp_idx_off: np.ndarray  # u64 [G+1]
    path_idx: np.ndarray  # u32 [NNZ]
    path_group_id: np.ndarray  # u32 [P]
    path_source_count: np.ndarray  # u32 [P]
    path_source_off: np.ndarray  # u64 [P+1]
    source_id: np.ndarray  # u32 [S]
    path_effective_length: np.ndarray  # f64 [P]

    _DTYPES = dict(
        cluster_row_off=np.uint64, cluster_path_off=np.uint64, row_count=np.uint32, row_noise=np.float64,
        row_grp_off=np.uint64, grp_prob=np.float64, grp_idx_off=np.uint64, path_idx=np.uint32,
        path_group_id=np.uint32, path_source_count=np.uint32, path_source_off=np.uint64, source_id=np.uint32,
        path_effective_length=np.float64)

    def __post_init__(self):
        for name, dt in self._DTYPES.items():
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=dt))

    @property
    def num_clusters(self) -> int:
        return len(self.cluster_row_off) - 1

    @property
    def num_rows(self) -> int:
        return int(self.cluster_row_off[-1])

    @property
    def num_paths(self) -> int:
        return int(self.cluster_path_off[-1])

    @property
    def total_reads(self) -> int:
        return int(self.row_count.astype(np.uint64).sum())

    def offsets32(self):
        """The two long offset arrays in 32 bits (rpvg_cluster_batch::row_grp_off32 / grp_idx_off32), made once and kept."""
        cached = getattr(self, "_offsets32", None)
        if cached is None:
            assert int(self.grp_idx_off[-1]) < 2 ** 32 and int(self.row_grp_off[-1]) < 2 ** 32
            cached = (np.ascontiguousarray(self.row_grp_off, dtype=np.uint32), np.ascontiguousarray(self.grp_idx_off, dtype=np.uint32))
            self._offsets32 = cached
        return cached

    def counts8(self):
        """The groups of every row and the paths of every group as one byte each (rpvg_cluster_batch::row_grp_count8 /
        grp_idx_count8), made once and kept; None when a count does not fit."""
        cached = getattr(self, "_counts8", False)
        if cached is False:
            rows, groups = np.diff(self.row_grp_off.astype(np.int64)), np.diff(self.grp_idx_off.astype(np.int64))
            fits = len(groups) > 0 and int(rows.max(initial=0)) <= 255 and int(groups.max(initial=0)) <= 255
            cached = (np.ascontiguousarray(rows, dtype=np.uint8), np.ascontiguousarray(groups, dtype=np.uint8)) if fits else None
            self._counts8 = cached
        return cached

    def narrow(self):
        """The narrow forms of three more arrays for the copy to the GPU (rpvg_cluster_batch::path_idx16, source_id16, row_count8 with
        the list of the rows whose count does not fit a byte), made once and kept: a dict with the arrays that fit."""
        cached = getattr(self, "_narrow", None)
        if cached is None:
            cached = {}
            paths = np.diff(self.cluster_path_off.astype(np.int64))
            if len(self.path_idx) and int(paths.max(initial=0)) < 65536:
                cached["path_idx16"] = np.ascontiguousarray(self.path_idx, dtype=np.uint16)
            if len(self.source_id) and int(self.source_id.max()) < 65536:
                cached["source_id16"] = np.ascontiguousarray(self.source_id, dtype=np.uint16)
            if len(self.row_count):
                rows = np.flatnonzero(self.row_count >= 255).astype(np.uint32)
                cached["row_count8"] = np.ascontiguousarray(np.minimum(self.row_count, 255), dtype=np.uint8)
                cached["row_count_escape_row"] = np.ascontiguousarray(rows)
                cached["row_count_escape_count"] = np.ascontiguousarray(self.row_count[rows], dtype=np.uint32)
                table, index = np.unique(self.row_noise, return_inverse=True)
                if len(table) <= 65536:
                    cached["row_noise16"] = np.ascontiguousarray(index, dtype=np.uint16)
                    cached["row_noise_table"] = np.ascontiguousarray(table, dtype=np.float64)
            self._narrow = cached
        return cached

    def as_c(self, compact: bool = False, narrow: bool = False) -> CClusterBatch:
        """compact: the 32-bit forms of the two long offset arrays instead of the 64-bit ones (a sixth fewer bytes to copy), and
        with them, where they fit, the counts of one byte that the copy to the GPU takes instead (a third fewer).  narrow: also
        the 16-bit path indices and source ids and the one-byte read counts, where they fit (narrow())."""
        # the returned struct borrows the arrays: keep `self` alive while it is in use
        if compact:
            row32, grp32 = self.offsets32()
            counts = self.counts8()
            tail = (_ptr(counts[0], u8p), _ptr(counts[1], u8p), len(self.grp_prob), len(self.path_idx)) if counts else (None, None, 0, 0)
            if narrow:
                forms = self.narrow()
                tail = tail + (_ptr(forms["path_idx16"], u16p) if "path_idx16" in forms else None,
                               _ptr(forms["source_id16"], u16p) if "source_id16" in forms else None,
                               _ptr(forms["row_count8"], u8p) if "row_count8" in forms else None,
                               _ptr(forms["row_count_escape_row"], u32p) if "row_count8" in forms else None,
                               _ptr(forms["row_count_escape_count"], u32p) if "row_count8" in forms else None,
                               len(forms["row_count_escape_row"]) if "row_count8" in forms else 0,
                               _ptr(forms["row_noise16"], u16p) if "row_noise16" in forms else None,
                               _ptr(forms["row_noise_table"], f64p) if "row_noise16" in forms else None,
                               len(forms["row_noise_table"]) if "row_noise16" in forms else 0)
            return CClusterBatch(
                self.num_clusters, _ptr(self.cluster_row_off, u64p), _ptr(self.cluster_path_off, u64p),
                _ptr(self.row_count, u32p), _ptr(self.row_noise, f64p), None,
                _ptr(self.grp_prob, f64p), None, _ptr(self.path_idx, u32p),
                _ptr(self.path_group_id, u32p), _ptr(self.path_source_count, u32p), _ptr(self.path_source_off, u64p),
                _ptr(self.source_id, u32p), _ptr(self.path_effective_length, f64p), _ptr(row32, u32p), _ptr(grp32, u32p), *tail)
        return CClusterBatch(
            self.num_clusters, _ptr(self.cluster_row_off, u64p), _ptr(self.cluster_path_off, u64p),
            _ptr(self.row_count, u32p), _ptr(self.row_noise, f64p), _ptr(self.row_grp_off, u64p),
            _ptr(self.grp_prob, f64p), _ptr(self.grp_idx_off, u64p), _ptr(self.path_idx, u32p),
            _ptr(self.path_group_id, u32p), _ptr(self.path_source_count, u32p), _ptr(self.path_source_off, u64p),
            _ptr(self.source_id, u32p), _ptr(self.path_effective_length, f64p), None, None, None, None, 0, 0)

    def cluster_range(self, first: int, last: int) -> "ClusterRange":
        """Clusters [first, last) of the batch as a batch of their own, without a copy of the rows (ClusterRange)."""
        return ClusterRange(self, first, last)

    # ---- construction from nested python data (tests, fixtures) -------------
    @staticmethod
    def from_clusters(clusters: Sequence[dict]) -> "ClusterBatch":
        """clusters: [{"paths": [{"group_id", "source_count", "source_ids", "effective_length"}...],
                       "rows": [(read_count, noise, [(prob, [idx...])...])...]}...]"""
        cro, cpo = [0], [0]
        rc, rn, rgo, gp, gio, pi = [], [], [0], [], [0], []
        pg, psc, pso, sid, pel = [], [], [0], [], []
        for cl in clusters:
            for p in cl["paths"]:
                pg.append(p.get("group_id", 0))
                psc.append(p.get("source_count", 1))
                sid.extend(p.get("source_ids", []))
                pso.append(len(sid))
                pel.append(p.get("effective_length", 0.0))
            cpo.append(len(pg))
            for (cnt, noise, groups) in cl["rows"]:
                rc.append(cnt)
                rn.append(noise)
                for (prob, idxs) in groups:
                    gp.append(prob)
                    pi.extend(idxs)
                    gio.append(len(pi))
                rgo.append(len(gp))
            cro.append(len(rc))
        return ClusterBatch(cro, cpo, rc, rn, rgo, gp, gio, pi, pg, psc, pso, sid, pel)

    def cluster(self, k: int) -> dict:
        """Inverse of from_clusters for one cluster."""
        paths = []
        for p in range(int(self.cluster_path_off[k]), int(self.cluster_path_off[k + 1])):
            paths.append(dict(group_id=int(self.path_group_id[p]), source_count=int(self.path_source_count[p]),
                              source_ids=[int(x) for x in self.source_id[int(self.path_source_off[p]):int(self.path_source_off[p + 1])]],
                              effective_length=float(self.path_effective_length[p])))
        rows = []
        for r in range(int(self.cluster_row_off[k]), int(self.cluster_row_off[k + 1])):
            groups = []
            for g in range(int(self.row_grp_off[r]), int(self.row_grp_off[r + 1])):
                groups.append((float(self.grp_prob[g]),
                               [int(x) for x in self.path_idx[int(self.grp_idx_off[g]):int(self.grp_idx_off[g + 1])]]))
            rows.append((int(self.row_count[r]), float(self.row_noise[r]), groups))
        return dict(paths=paths, rows=rows)

    @staticmethod
    def concat(batches: Sequence["ClusterBatch"]) -> "ClusterBatch":
        """The clusters of several batches back to back."""
        def offs(name):
            parts, base = [np.zeros(1, dtype=np.uint64)], 0
            for b in batches:
                o = getattr(b, name).astype(np.uint64)
                parts.append(o[1:] + np.uint64(base))
                base += int(o[-1])
            return np.concatenate(parts)

        def cat(name):
            return np.concatenate([getattr(b, name) for b in batches])

        return ClusterBatch(
            offs("cluster_row_off"), offs("cluster_path_off"), cat("row_count"), cat("row_noise"), offs("row_grp_off"),
            cat("grp_prob"), offs("grp_idx_off"), cat("path_idx"), cat("path_group_id"), cat("path_source_count"),
            offs("path_source_off"), cat("source_id"), cat("path_effective_length"))

    def select(self, ks: Sequence[int]) -> "ClusterBatch":
        """Sub-batch holding clusters ks (in that order); used to shard a batch across ranks."""
        ks = np.asarray(ks, dtype=np.int64)
        r0, r1 = self.cluster_row_off[ks].astype(np.int64), self.cluster_row_off[ks + 1].astype(np.int64)
        p0, p1 = self.cluster_path_off[ks].astype(np.int64), self.cluster_path_off[ks + 1].astype(np.int64)

        def ranges(lo, hi):
            n = hi - lo
            tot = int(n.sum())
            if tot == 0:
                return np.zeros(0, dtype=np.int64)
            starts = np.repeat(lo - np.concatenate(([0], np.cumsum(n)[:-1])), n)
            return starts + np.arange(tot, dtype=np.int64)

        rows = ranges(r0, r1)
        paths = ranges(p0, p1)
        g0, g1 = self.row_grp_off[rows].astype(np.int64), self.row_grp_off[rows + 1].astype(np.int64)
        grps = ranges(g0, g1)
        e0, e1 = self.grp_idx_off[grps].astype(np.int64), self.grp_idx_off[grps + 1].astype(np.int64)
        ents = ranges(e0, e1)
        s0, s1 = self.path_source_off[paths].astype(np.int64), self.path_source_off[paths + 1].astype(np.int64)
        srcs = ranges(s0, s1)

        def offs(n):
            return np.concatenate(([0], np.cumsum(n))).astype(np.uint64)

        return ClusterBatch(
            offs(r1 - r0), offs(p1 - p0), self.row_count[rows], self.row_noise[rows], offs(g1 - g0),
            self.grp_prob[grps], offs(e1 - e0), self.path_idx[ents], self.path_group_id[paths],
            self.path_source_count[paths], offs(s1 - s0), self.source_id[srcs], self.path_effective_length[paths])


class ClusterRange:
    """Clusters [first, last) of a ClusterBatch as an rpvg_cluster_batch of their own: the long arrays are views of the parent's
    (rows, groups and entries of consecutive clusters are consecutive), the two long offset arrays travel as the parent's counts of
    one byte (a slice of counts is the counts of the slice), and only the small per-cluster and per-path offset arrays are made
    anew.  What cuts one data set into parts for the batch pipeline without touching its 190 MB."""

    def __init__(self, parent: ClusterBatch, first: int, last: int):
        counts = parent.counts8()
        if counts is None:
            raise ValueError("a cluster range needs the parent's counts of one byte (no row with more than 255 groups, no group with more than 255 paths)")
        self.parent, self.first, self.last = parent, first, last
        r0, r1 = int(parent.cluster_row_off[first]), int(parent.cluster_row_off[last])
        p0, p1 = int(parent.cluster_path_off[first]), int(parent.cluster_path_off[last])
        g0, g1 = int(parent.row_grp_off[r0]), int(parent.row_grp_off[r1])
        e0, e1 = int(parent.grp_idx_off[g0]), int(parent.grp_idx_off[g1])
        s0, s1 = int(parent.path_source_off[p0]), int(parent.path_source_off[p1])
        self.cluster_row_off = np.ascontiguousarray(parent.cluster_row_off[first:last + 1] - np.uint64(r0), dtype=np.uint64)
        self.cluster_path_off = np.ascontiguousarray(parent.cluster_path_off[first:last + 1] - np.uint64(p0), dtype=np.uint64)
        self.path_source_off = np.ascontiguousarray(parent.path_source_off[p0:p1 + 1] - np.uint64(s0), dtype=np.uint64)
        self.row_count, self.row_noise = parent.row_count[r0:r1], parent.row_noise[r0:r1]
        self.row_grp_count8, self.grp_idx_count8 = counts[0][r0:r1], counts[1][g0:g1]
        self.grp_prob, self.path_idx = parent.grp_prob[g0:g1], parent.path_idx[e0:e1]
        self.path_group_id, self.path_source_count = parent.path_group_id[p0:p1], parent.path_source_count[p0:p1]
        self.source_id, self.path_effective_length = parent.source_id[s0:s1], parent.path_effective_length[p0:p1]
        self.num_clusters, self.num_rows = last - first, r1 - r0
        self.total_reads = int(self.row_count.astype(np.uint64).sum())
        # the parent's narrow forms (ClusterBatch.narrow): slices again, the listed rows moved to the range's numbering
        self.narrow_forms = {}
        forms = parent.narrow()
        if "path_idx16" in forms:
            self.narrow_forms["path_idx16"] = forms["path_idx16"][e0:e1]
        if "source_id16" in forms:
            self.narrow_forms["source_id16"] = forms["source_id16"][s0:s1]
        if "row_count8" in forms:
            listed = forms["row_count_escape_row"]
            a, b = int(np.searchsorted(listed, r0)), int(np.searchsorted(listed, r1))
            self.narrow_forms["row_count8"] = forms["row_count8"][r0:r1]
            self.narrow_forms["row_count_escape_row"] = np.ascontiguousarray(listed[a:b] - np.uint32(r0), dtype=np.uint32)
            self.narrow_forms["row_count_escape_count"] = forms["row_count_escape_count"][a:b]
        if "row_noise16" in forms:  # (the whole table with every range: a few kilobytes)
            self.narrow_forms["row_noise16"] = forms["row_noise16"][r0:r1]
            self.narrow_forms["row_noise_table"] = forms["row_noise_table"]

    def as_c(self, compact: bool = True, narrow: bool = False) -> CClusterBatch:
        assert compact, "a cluster range has no offset arrays of its own"
        tail = ()
        if narrow:
            forms = self.narrow_forms
            tail = (_ptr(forms["path_idx16"], u16p) if "path_idx16" in forms else None,
                    _ptr(forms["source_id16"], u16p) if "source_id16" in forms else None,
                    _ptr(forms["row_count8"], u8p) if "row_count8" in forms else None,
                    _ptr(forms["row_count_escape_row"], u32p) if "row_count8" in forms else None,
                    _ptr(forms["row_count_escape_count"], u32p) if "row_count8" in forms else None,
                    len(forms["row_count_escape_row"]) if "row_count8" in forms else 0,
                    _ptr(forms["row_noise16"], u16p) if "row_noise16" in forms else None,
                    _ptr(forms["row_noise_table"], f64p) if "row_noise16" in forms else None,
                    len(forms["row_noise_table"]) if "row_noise16" in forms else 0)
        return CClusterBatch(
            self.num_clusters, _ptr(self.cluster_row_off, u64p), _ptr(self.cluster_path_off, u64p),
            _ptr(self.row_count, u32p), _ptr(self.row_noise, f64p), None,
            _ptr(self.grp_prob, f64p), None, _ptr(self.path_idx, u32p),
            _ptr(self.path_group_id, u32p), _ptr(self.path_source_count, u32p), _ptr(self.path_source_off, u64p),
            _ptr(self.source_id, u32p), _ptr(self.path_effective_length, f64p), None, None,
            _ptr(self.row_grp_count8, u8p), _ptr(self.grp_idx_count8, u8p), len(self.grp_prob), len(self.path_idx), *tail)


@dataclass
class ClusterEstimates:
    """Decoded PathClusterEstimates of one cluster (src/path_cluster_estimates.hpp:49-57)."""

    path_group_sets: List[Tuple[int, ...]]
    posteriors: np.ndarray
    abundances: np.ndarray
    noise_count: float
    total_count: float
    em_iters: List[int] = field(default_factory=list)
    em_cols: List[Tuple[int, ...]] = field(default_factory=list)
    # CountSamples of the cluster (-n > 0): [(path_ids, noise_samples[n], abundance_samples[n, len(path_ids)])...]
    gibbs_samples: List[Tuple[Tuple[int, ...], np.ndarray, np.ndarray]] = field(default_factory=list)

    def keyed(self) -> Dict[Tuple[int, ...], Tuple[float, Tuple[float, ...]]]:
        """{group set -> (posterior, abundances of its members)} — order-free view.

        `transcripts`/`strains` carry one abundance per set, `haplotype-transcripts` one per member,
        `haplotypes` none."""
        out = {}
        n_sets = len(self.path_group_sets)
        n_members = sum(len(s) for s in self.path_group_sets)
        a = 0
        for i, s in enumerate(self.path_group_sets):
            if len(self.abundances) == 0:
                ab = ()
            elif len(self.abundances) == n_sets and n_members != n_sets:
                ab = (float(self.abundances[i]),)
            else:
                ab = tuple(float(x) for x in self.abundances[a:a + len(s)])
                a += len(s)
            assert s not in out, "duplicate group set"
            out[s] = (float(self.posteriors[i]), ab)
        return out


def decode_view(view: CEstimatesView) -> List[ClusterEstimates]:
    K = view.num_clusters

    def arr(ptr, n, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

    set_off = arr(view.set_off, K + 1, np.uint64).astype(np.int64)
    S = int(set_off[-1])
    member_off = arr(view.member_off, S + 1, np.uint64).astype(np.int64)
    members = arr(view.members, int(member_off[-1]), np.uint32)
    post = arr(view.posteriors, S, np.float64)
    abund_off = arr(view.abund_off, K + 1, np.uint64).astype(np.int64)
    abund = arr(view.abundances, int(abund_off[-1]), np.float64)
    noise = arr(view.noise_count, K, np.float64)
    total = arr(view.total_count, K, np.float64)
    em_off = arr(view.em_off, K + 1, np.uint64).astype(np.int64)
    E = int(em_off[-1])
    em_iters = arr(view.em_iters, E, np.uint32)
    em_col_off = arr(view.em_col_off, E + 1, np.uint64).astype(np.int64)
    em_cols = arr(view.em_cols, int(em_col_off[-1]), np.uint32)
    g_off = arr(view.gibbs_off, K + 1, np.uint64).astype(np.int64)
    Gs = int(g_off[-1])
    gp_off = arr(view.gibbs_path_off, Gs + 1, np.uint64).astype(np.int64)
    gn_off = arr(view.gibbs_noise_off, Gs + 1, np.uint64).astype(np.int64)
    ga_off = arr(view.gibbs_abund_off, Gs + 1, np.uint64).astype(np.int64)
    g_path = arr(view.gibbs_path, int(gp_off[-1]), np.uint32)
    g_noise = arr(view.gibbs_noise, int(gn_off[-1]), np.float64)
    g_abund = arr(view.gibbs_abund, int(ga_off[-1]), np.float64)
    out = []
    for k in range(K):
        sets = [tuple(int(x) for x in members[member_off[s]:member_off[s + 1]]) for s in range(set_off[k], set_off[k + 1])]
        gibbs = []
        for g in range(g_off[k], g_off[k + 1]):
            ids = tuple(int(x) for x in g_path[gp_off[g]:gp_off[g + 1]])
            ns = g_noise[gn_off[g]:gn_off[g + 1]].copy()
            ab = g_abund[ga_off[g]:ga_off[g + 1]].copy().reshape(len(ns), len(ids)) if len(ids) else np.zeros((len(ns), 0))
            gibbs.append((ids, ns, ab))
        out.append(ClusterEstimates(
            sets, post[set_off[k]:set_off[k + 1]].copy(), abund[abund_off[k]:abund_off[k + 1]].copy(),
            float(noise[k]), float(total[k]),
            [int(x) for x in em_iters[em_off[k]:em_off[k + 1]]],
            [tuple(int(x) for x in em_cols[em_col_off[e]:em_col_off[e + 1]]) for e in range(em_off[k], em_off[k + 1])],
            gibbs))
    return out
