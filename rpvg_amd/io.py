"""ctypes binding of the file formats either side of the hot path and of the replay driver
(``rpvg_amd/host/io``): `--write-probs` dumps, `-f` path info, the reference's result TSVs."""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import engine as _engine, hip, synth
from .batch import CClusterBatch, CEstimatesView, CParams, ClusterBatch


def _lib():
    L = _engine.lib()
    L.rpvg_amd_io_last_error.restype = C.c_char_p
    L.rpvg_amd_batch_write_files.restype = C.c_int
    L.rpvg_amd_batch_write_files.argtypes = [C.POINTER(CClusterBatch), C.c_char_p, C.c_char_p, C.c_double]
    L.rpvg_amd_batch_write_files_ranked.restype = C.c_int
    L.rpvg_amd_batch_write_files_ranked.argtypes = [C.POINTER(CClusterBatch), C.c_char_p, C.c_char_p, C.c_double, C.c_void_p, C.c_void_p]
    L.rpvg_amd_batch_read_files.restype = C.c_void_p
    L.rpvg_amd_batch_read_files.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_double]
    L.rpvg_amd_info_table.restype = C.c_int64
    L.rpvg_amd_info_table.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    L.rpvg_amd_replay.restype = C.c_int64
    L.rpvg_amd_replay.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(CParams), C.c_char_p, C.c_int, C.c_uint32]
    L.rpvg_amd_write_estimates.restype = C.c_int
    L.rpvg_amd_write_estimates.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(CParams), C.POINTER(CEstimatesView),
                                           C.c_char_p, C.c_uint32]
    return L


def _fail(what):
    raise hip.EngineError(f"{what} failed: {_lib().rpvg_amd_io_last_error().decode()}")


def info_table(info_path: str, parse_haplotype_ids: bool = True, use_transcript_names: bool = False):
    """The `-f` parser's view of a path info file: [(key, name, group_id, source_count, [source ids ascending])], by key."""
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "table.tsv")
        n = _lib().rpvg_amd_info_table(info_path.encode(), 1 if parse_haplotype_ids else 0, 1 if use_transcript_names else 0, out.encode())
        if n < 0:
            _fail("info_table")
        rows = []
        for line in open(out):
            key, name, group, count, ids = line.rstrip("\n").split("\t")
            rows.append((key, name, int(group), int(count), [int(x) for x in ids.split(",")] if ids else []))
    assert len(rows) == n
    return rows


def write_batch_files(batch: ClusterBatch, probs_path: str, info_path: str, prob_precision: float = 1e-8, num_align_lists=None,
                      cluster_index=None):
    """The batch as a `--write-probs` dump plus a matching `-f` path info TSV (generated names).  With num_align_lists and
    cluster_index (one per cluster) every block carries the rank key of the reference's cluster loop (src/main.cpp:811-827)
    in its marker line — the replay then numbers and seeds the clusters as the run that produced them did."""
    import numpy as np
    cb = batch.as_c()
    if num_align_lists is None:
        if _lib().rpvg_amd_batch_write_files(C.byref(cb), probs_path.encode(), info_path.encode(), prob_precision) != 0:
            _fail("write_batch_files")
        return
    lists = np.ascontiguousarray(num_align_lists, dtype=np.uint64)
    index = np.ascontiguousarray(cluster_index, dtype=np.uint64)
    assert len(lists) == batch.num_clusters == len(index)
    if _lib().rpvg_amd_batch_write_files_ranked(C.byref(cb), probs_path.encode(), info_path.encode(), prob_precision,
                                                C.c_void_p(lists.ctypes.data), C.c_void_p(index.ctypes.data)) != 0:
        _fail("write_batch_files")


def read_batch_files(probs_path: str, info_path: Optional[str], parse_haplotype_ids: bool = True,
                     prob_precision: float = 1e-8) -> ClusterBatch:
    """A dump (+ path info) as a flat batch, clusters ranked as the replay ranks them: by the rank key of the blocks when every
    block has one, else by read count."""
    L = _lib()
    synth._bind()
    h = L.rpvg_amd_batch_read_files(probs_path.encode(), (info_path or "").encode(), 1 if parse_haplotype_ids else 0, prob_precision)
    if not h:
        _fail("read_batch_files")
    try:
        return synth._to_batch(L, h)
    finally:
        L.rpvg_amd_synth_free(h)


def replay(probs_path: str, info_path: Optional[str], model: str, params: CParams, prefix: str, device: int = 0,
           unaligned_read_count: int = 0) -> int:
    """Dump -> GPU estimators -> the reference's result files.  Returns the number of clusters."""
    n = _lib().rpvg_amd_replay(probs_path.encode(), (info_path or "").encode(), model.encode(), C.byref(params), prefix.encode(),
                               device, unaligned_read_count)
    if n < 0:
        _fail("replay")
    return int(n)


def write_estimates(probs_path: str, info_path: Optional[str], model: str, params: CParams, view: CEstimatesView, prefix: str,
                    unaligned_read_count: int = 0):
    """The reference's result files for estimates computed elsewhere (writers only; no GPU)."""
    if _lib().rpvg_amd_write_estimates(probs_path.encode(), (info_path or "").encode(), model.encode(), C.byref(params),
                                       C.byref(view), prefix.encode(), unaligned_read_count) != 0:
        _fail("write_estimates")
