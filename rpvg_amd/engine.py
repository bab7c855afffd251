"""ctypes binding of the harness entry points (librpvg_amd_harness.so) over the host-side estimator classes
(librpvg_amd_host.so).

The work happens in C++ (``rpvg_amd/host``: PathEstimator classes over the C
ABI of the HIP engine); this module only marshals flat batches in and
estimates out for tests and bench.  No fallback: a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

from . import hip
from .batch import CClusterBatch, CEstimatesView, CParams, ClusterBatch, ClusterEstimates, decode_view

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "librpvg_amd_harness.so")  # links librpvg_amd_host.so, the product

MODELS = ("transcripts", "strains", "haplotype-transcripts", "haplotypes")

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise hip.EngineError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (there is no CPU fallback)")
        hip.lib()  # librpvg_hip.so first (also resolved through the rpath)
        L = C.CDLL(LIB_PATH)
        L.rpvg_amd_last_error.restype = C.c_char_p
        L.rpvg_amd_engine_create.restype = C.c_void_p
        L.rpvg_amd_engine_create.argtypes = [C.c_int]
        L.rpvg_amd_engine_create_uploader.restype = C.c_void_p
        L.rpvg_amd_engine_create_uploader.argtypes = [C.c_int]
        L.rpvg_amd_engine_destroy.argtypes = [C.c_void_p]
        L.rpvg_amd_engine_ctx.restype = C.c_void_p
        L.rpvg_amd_engine_ctx.argtypes = [C.c_void_p]
        L.rpvg_amd_engine_stats_get.argtypes = [C.c_void_p, C.c_void_p]
        L.rpvg_amd_engine_stats_reset.argtypes = [C.c_void_p]
        L.rpvg_amd_host_threads.restype = C.c_int
        L.rpvg_amd_batch_prepare.restype = C.c_void_p
        L.rpvg_amd_batch_prepare.argtypes = [C.c_void_p, C.POINTER(CClusterBatch), C.c_int]
        L.rpvg_amd_batch_free.argtypes = [C.c_void_p]
        L.rpvg_amd_batch_reupload.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CClusterBatch), C.POINTER(C.c_double)]
        L.rpvg_amd_batch_prepare_from_alignments.restype = C.c_void_p
        L.rpvg_amd_batch_prepare_from_alignments.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CClusterBatch), C.c_double, C.c_double,
                                                             C.c_double, C.c_uint32, C.c_int, C.c_double, C.c_double,
                                                             C.POINTER(C.c_double)]
        L.rpvg_amd_batch_prepare_synth_dense.restype = C.c_void_p
        L.rpvg_amd_batch_prepare_synth_dense.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32]
        L.rpvg_amd_run.restype = C.c_void_p
        L.rpvg_amd_run.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(CParams), C.POINTER(C.c_double)]
        L.rpvg_amd_run_team.restype = C.c_int
        L.rpvg_amd_run_team.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(CParams), C.c_int, C.POINTER(C.c_double)]
        L.rpvg_amd_run_team_result.restype = C.c_void_p
        L.rpvg_amd_run_team_result.argtypes = [C.c_void_p]
        L.rpvg_amd_run_inplace.restype = C.c_int
        L.rpvg_amd_run_inplace.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(CParams), C.POINTER(C.c_double)]
        L.rpvg_amd_run_from_alignments_inplace.restype = C.c_int
        L.rpvg_amd_run_from_alignments_inplace.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(CParams), C.POINTER(C.c_double),
                                                           C.POINTER(C.c_double)]
        L.rpvg_amd_group_create.restype = C.c_void_p
        L.rpvg_amd_group_create.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.rpvg_amd_group_destroy.argtypes = [C.c_void_p]
        L.rpvg_amd_group_has_communicator.argtypes = [C.c_void_p]
        L.rpvg_amd_group_run.restype = C.c_void_p
        L.rpvg_amd_group_run.argtypes = [C.c_void_p, C.POINTER(CClusterBatch), C.c_char_p, C.POINTER(CParams), C.POINTER(C.c_double)]
        L.rpvg_amd_group_partition.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.rpvg_amd_group_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.rpvg_amd_pipeline_create.restype = C.c_void_p
        L.rpvg_amd_pipeline_create.argtypes = [C.c_int, C.c_char_p, C.POINTER(CParams), C.c_int]
        L.rpvg_amd_pipeline_destroy.argtypes = [C.c_void_p]
        L.rpvg_amd_pipeline_workers.argtypes = [C.c_void_p]
        L.rpvg_amd_pipeline_prepare_slots.argtypes = [C.c_void_p, C.POINTER(CClusterBatch), C.c_int]
        L.rpvg_amd_pipeline_submit.argtypes = [C.c_void_p, C.POINTER(CClusterBatch), C.c_int]
        L.rpvg_amd_pipeline_prepare_slot.argtypes = [C.c_void_p, C.POINTER(CClusterBatch), C.c_int]
        L.rpvg_amd_pipeline_wait.argtypes = [C.c_void_p]
        L.rpvg_amd_pipeline_result.restype = C.c_void_p
        L.rpvg_amd_pipeline_result.argtypes = [C.c_void_p, C.c_int]
        L.rpvg_amd_pipeline_stats_get.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.rpvg_amd_pipeline_stats_reset.argtypes = [C.c_void_p]
        L.rpvg_amd_pipeline_completions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.rpvg_amd_result_view.argtypes = [C.c_void_p, C.POINTER(CEstimatesView)]
        L.rpvg_amd_result_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _err() -> str:
    return lib().rpvg_amd_last_error().decode()


class Engine:
    """One GPU (HipEngine) shared by the estimators created on it."""

    def __init__(self, device: int = 0, uploader: bool = False):
        """uploader: an engine that only uploads batches (PreparedBatch.reupload) next to the one that estimates."""
        self.handle = lib().rpvg_amd_engine_create_uploader(device) if uploader else lib().rpvg_amd_engine_create(device)
        if not self.handle:
            raise hip.EngineError(f"engine create failed: {_err()}")

    def close(self):
        if self.handle:
            lib().rpvg_amd_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ctx(self):
        return C.c_void_p(lib().rpvg_amd_engine_ctx(self.handle))

    def stats(self) -> dict:
        s = hip.CKernelStats()
        if lib().rpvg_amd_engine_stats_get(self.handle, C.byref(s)) != 0:
            raise hip.EngineError(f"stats failed: {_err()}")
        return s.as_dict()

    def reset_stats(self):
        if lib().rpvg_amd_engine_stats_reset(self.handle) != 0:
            raise hip.EngineError(f"stats reset failed: {_err()}")

    def info(self) -> Tuple[str, int, int]:
        name = C.create_string_buffer(256)
        cus, mem = C.c_uint32(0), C.c_uint64(0)
        hip._check(hip.lib().rpvg_hip_device_info(self._ctx(), name, 256, C.byref(cus), C.byref(mem)), "rpvg_hip_device_info")
        return name.value.decode(), cus.value, mem.value

    def prepare(self, batch: ClusterBatch, per_cluster: bool = False) -> "PreparedBatch":
        return PreparedBatch(self, batch, per_cluster)

    def prepare_from_alignments(self, alignments, path_info: ClusterBatch, frag=(300.0, 50.0, 0.0, 10), is_single_end: bool = False,
                                min_noise_prob: float = 1e-4, prob_precision: float = 1e-8) -> "PreparedBatch":
        """Batch whose rows are constructed on the GPU from alignment-path lists (rpvg_amd/host/read_rows.hpp):
        alignments = rows.AlignmentBatch, path_info = a ClusterBatch whose path arrays describe the clusters' paths,
        frag = (loc, scale, shape, sd_max_multi) of the FragmentLengthDist."""
        prep = PreparedBatch.__new__(PreparedBatch)
        prep.engine = self
        prep.batch = path_info
        ca, cb = alignments.as_c(), path_info.as_c()
        secs = C.c_double(0)
        prep.handle = lib().rpvg_amd_batch_prepare_from_alignments(
            self.handle, C.byref(ca), C.byref(cb), frag[0], frag[1], frag[2], int(frag[3]), 1 if is_single_end else 0,
            min_noise_prob, prob_precision, C.byref(secs))
        if not prep.handle:
            raise hip.EngineError(f"batch prepare from alignments failed: {_err()}")
        prep.row_construction_seconds = secs.value
        return prep

    def prepare_synth_dense(self, seed: int, rows: int, paths: int) -> "PreparedBatch":
        """BASELINE.json configs[1] as a resident batch for the estimator classes: one cluster of `rows` rows over all `paths`
        paths, generated on the GPU (rpvg_hip_synth_dense_cluster_batch; every row one read pair)."""
        prep = PreparedBatch.__new__(PreparedBatch)
        prep.engine = self
        prep.batch = None
        prep.handle = lib().rpvg_amd_batch_prepare_synth_dense(self.handle, C.c_uint64(seed), C.c_uint64(rows), C.c_uint32(paths))
        if not prep.handle:
            raise hip.EngineError(f"synthetic dense batch failed: {_err()}")
        return prep

    def run(self, model: str, params: CParams, prepared: "PreparedBatch") -> Tuple[List[ClusterEstimates], float]:
        """Estimates of every cluster + wall seconds of the estimator call (inputs already on the GPU)."""
        secs = C.c_double(0)
        h = lib().rpvg_amd_run(self.handle, prepared.handle, model.encode(), C.byref(params), C.byref(secs))
        if not h:
            raise hip.EngineError(f"run({model}) failed: {_err()}")
        try:
            view = CEstimatesView()
            lib().rpvg_amd_result_view(h, C.byref(view))
            out = decode_view(view)
        finally:
            lib().rpvg_amd_result_free(h)
        return out, secs.value

    def run_team(self, model: str, params: CParams, prepared: "PreparedBatch", threads: int, decode: bool = True):
        """The reference's cluster loop (src/main.cpp:829,976-977): PathEstimator::estimate() once per cluster from an OpenMP team
        of `threads`; `prepared` was made with per_cluster=True.  Returns (estimates or None, wall seconds)."""
        secs = C.c_double(0)
        if lib().rpvg_amd_run_team(self.handle, prepared.handle, model.encode(), C.byref(params), threads, C.byref(secs)) != 0:
            raise hip.EngineError(f"run_team({model}) failed: {_err()}")
        if not decode:
            return None, secs.value
        h = lib().rpvg_amd_run_team_result(prepared.handle)
        if not h:
            raise hip.EngineError(f"run_team({model}) result failed: {_err()}")
        try:
            view = CEstimatesView()
            lib().rpvg_amd_result_view(h, C.byref(view))
            out = decode_view(view)
        finally:
            lib().rpvg_amd_result_free(h)
        return out, secs.value

    def run_raw(self, model: str, params: CParams, prepared: "PreparedBatch") -> float:
        """Like run() but leaves the estimates in the C++ PathClusterEstimates containers of the prepared
        batch instead of flattening and decoding them (bench inner loop)."""
        secs = C.c_double(0)
        rc = lib().rpvg_amd_run_inplace(self.handle, prepared.handle, model.encode(), C.byref(params), C.byref(secs))
        if rc != 0:
            raise hip.EngineError(f"run({model}) failed: {_err()}")
        return secs.value


    def run_from_alignments_raw(self, model: str, params: CParams, prepared: "PreparedBatch") -> Tuple[float, float]:
        """One pass alignment-path lists (resident) -> rows -> estimates, everything on the GPU; returns the wall seconds
        of the row construction and of the estimator call.  `prepared` comes from prepare_from_alignments()."""
        rows_s, est_s = C.c_double(0), C.c_double(0)
        rc = lib().rpvg_amd_run_from_alignments_inplace(self.handle, prepared.handle, model.encode(), C.byref(params),
                                                        C.byref(rows_s), C.byref(est_s))
        if rc != 0:
            raise hip.EngineError(f"run_from_alignments({model}) failed: {_err()}")
        return rows_s.value, est_s.value


class DeviceGroup:
    """The GPUs of one node behind one call (rpvg_amd/host/device_group.hpp): one engine and one host thread per
    entry of `devices`, clusters bin-packed over them, final abundance gather over the group's communicator."""

    def __init__(self, devices):
        arr = (C.c_int * len(devices))(*devices)
        self.handle = lib().rpvg_amd_group_create(arr, len(devices))
        if not self.handle:
            raise hip.EngineError(f"device group create failed: {_err()}")
        self.num_clusters = 0

    def close(self):
        if self.handle:
            lib().rpvg_amd_group_destroy(self.handle)
            self.handle = None

    def has_communicator(self) -> bool:
        return bool(lib().rpvg_amd_group_has_communicator(self.handle))

    def run(self, model: str, params: CParams, batch: ClusterBatch) -> Tuple[List[ClusterEstimates], float]:
        secs = C.c_double(0)
        cb = batch.as_c()
        h = lib().rpvg_amd_group_run(self.handle, C.byref(cb), model.encode(), C.byref(params), C.byref(secs))
        if not h:
            raise hip.EngineError(f"group run({model}) failed: {_err()}")
        try:
            view = CEstimatesView()
            lib().rpvg_amd_result_view(h, C.byref(view))
            out = decode_view(view)
        finally:
            lib().rpvg_amd_result_free(h)
        self.num_clusters = batch.num_clusters
        return out, secs.value

    def partition(self):
        import numpy as np
        out = np.zeros(self.num_clusters, dtype=np.uint32)
        if lib().rpvg_amd_group_partition(self.handle, C.c_void_p(out.ctypes.data), self.num_clusters) != 0:
            raise hip.EngineError(f"group partition failed: {_err()}")
        return out

    def gather(self, capacity: int):
        """(abundances of all clusters in cluster order, TPM denominator) after the collective of the last run."""
        import numpy as np
        out = np.zeros(max(1, capacity), dtype=np.float64)
        count, total = C.c_uint64(0), C.c_double(0)
        if lib().rpvg_amd_group_gather(self.handle, C.c_void_p(out.ctypes.data), capacity, C.byref(count), C.byref(total)) != 0:
            raise hip.EngineError(f"group gather failed: {_err()}")
        return out[:count.value].copy(), total.value


class Pipeline:
    """Several batches in flight on one GPU (rpvg_amd/host/batch_pipeline.hpp): an uploader thread and `workers` estimator
    threads, each with an engine of its own; submit() returns at once, wait() when every submitted batch is done."""

    def __init__(self, model: str, params: CParams, device: int = 0, workers: int = 0):
        self.handle = lib().rpvg_amd_pipeline_create(device, model.encode(), C.byref(params), workers)
        if not self.handle:
            raise hip.EngineError(f"pipeline create failed: {_err()}")
        self._keep = []  # the host batches (and their C views) of the submissions since the last wait()

    @property
    def workers(self) -> int:
        return int(lib().rpvg_amd_pipeline_workers(self.handle))

    def prepare_slots(self, batch: ClusterBatch, slots: int):
        """`slots` sets of estimates containers for batches with the clusters (paths) of `batch`."""
        cb = batch.as_c()
        if lib().rpvg_amd_pipeline_prepare_slots(self.handle, C.byref(cb), slots) != 0:
            raise hip.EngineError(f"pipeline prepare failed: {_err()}")

    def prepare_slot(self, slot: int, part):
        """The containers of one slot for the clusters of `part` (a ClusterBatch or a ClusterRange of one): the parts of one data
        set go to slots of their own."""
        cb = part.as_c(True)
        if lib().rpvg_amd_pipeline_prepare_slot(self.handle, C.byref(cb), slot) != 0:
            raise hip.EngineError(f"pipeline prepare failed: {_err()}")

    def submit(self, batch, slot: int, compact: bool = False, narrow: bool = False):
        """compact / narrow: the forms of the batch's arrays made for the copy (ClusterBatch.as_c).  `batch`: a ClusterBatch or a
        ClusterRange."""
        cb = batch.as_c(compact, narrow) if narrow else batch.as_c(compact)
        self._keep.append((batch, cb))
        if lib().rpvg_amd_pipeline_submit(self.handle, C.byref(cb), slot) != 0:
            raise hip.EngineError(f"pipeline submit failed: {_err()}")

    def wait(self):
        rc = lib().rpvg_amd_pipeline_wait(self.handle)
        self._keep.clear()
        if rc != 0:
            raise hip.EngineError(f"pipeline batch failed: {_err()}")

    def result(self, slot: int) -> List[ClusterEstimates]:
        h = lib().rpvg_amd_pipeline_result(self.handle, slot)
        if not h:
            raise hip.EngineError(f"pipeline result failed: {_err()}")
        try:
            view = CEstimatesView()
            lib().rpvg_amd_result_view(h, C.byref(view))
            return decode_view(view)
        finally:
            lib().rpvg_amd_result_free(h)

    def stats(self) -> dict:
        s = hip.CKernelStats()
        up = (C.c_double * 6)()
        if lib().rpvg_amd_pipeline_stats_get(self.handle, C.byref(s), up) != 0:
            raise hip.EngineError(f"pipeline stats failed: {_err()}")
        out = s.as_dict()
        out["upload_seconds_per_batch"] = up[0]
        out["upload_copies_ms_per_batch"] = up[1]   # HIP-event span around the H2D copies of a batch
        out["upload_kernels_ms_per_batch"] = up[2]  # ... around the kernels behind them
        out["worker_finish_ms_per_batch"], out["worker_estimate_ms_per_batch"], out["worker_idle_ms_per_batch"] = up[3] * 1e3, up[4] * 1e3, up[5] * 1e3
        return out

    def reset_stats(self):
        if lib().rpvg_amd_pipeline_stats_reset(self.handle) != 0:
            raise hip.EngineError(f"pipeline stats reset failed: {_err()}")

    def completions(self, capacity: int = 4096):
        """Seconds since the last reset_stats() at which the batches since then were done, in order of completion."""
        import numpy as np
        out = np.zeros(capacity, dtype=np.float64)
        count = C.c_uint64(0)
        lib().rpvg_amd_pipeline_completions(self.handle, C.c_void_p(out.ctypes.data), capacity, C.byref(count))
        return out[:min(capacity, count.value)].copy()

    def close(self):
        if self.handle:
            lib().rpvg_amd_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PreparedBatch:
    def __init__(self, engine: Engine, batch: ClusterBatch, per_cluster: bool):
        self.engine = engine
        self.batch = batch
        cb = batch.as_c()
        self.handle = lib().rpvg_amd_batch_prepare(engine.handle, C.byref(cb), 1 if per_cluster else 0)
        if not self.handle:
            raise hip.EngineError(f"batch prepare failed: {_err()}")

    def reupload(self, engine: "Engine", batch: Optional[ClusterBatch] = None, compact: bool = False) -> float:
        """Replaces the resident rows by a fresh upload of `batch` (default: the batch this was prepared from) through
        `engine` (an uploader engine on the same GPU keeps the copy off the estimating engine's streams); seconds."""
        cb = (batch or self.batch).as_c(compact)
        secs = C.c_double(0)
        if lib().rpvg_amd_batch_reupload(engine.handle, self.handle, C.byref(cb), C.byref(secs)) != 0:
            raise hip.EngineError(f"batch reupload failed: {_err()}")
        return secs.value

    def free(self):
        if self.handle:
            lib().rpvg_amd_batch_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def run(model: str, params: CParams, batch: ClusterBatch, device: int = 0, per_cluster: bool = False) -> List[ClusterEstimates]:
    """One-shot convenience: upload, estimate, decode."""
    eng = Engine(device)
    try:
        prep = eng.prepare(batch, per_cluster)
        try:
            out, _ = eng.run(model, params, prep)
        finally:
            prep.free()
    finally:
        eng.close()
    return out
