"""Cluster sharding across GPUs (one process per GPU) and the final result gather.

Path clusters are independent units of inference (the reference schedules them dynamically over OpenMP
threads, src/main.cpp:829), so the multi-GPU path has no data-path collective: every rank runs the
estimators on its own clusters and the per-cluster results are gathered once at the end
(``torch.distributed`` all_gather: RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from .batch import ClusterBatch


def cluster_costs(batch: ClusterBatch) -> np.ndarray:
    """Cost proxy of a cluster: entries + rows * (paths + 1) — the size of its dense probability matrix plus
    its sparse rows (the reference orders clusters by their number of alignments, src/main.cpp:811-827)."""
    rows = np.diff(batch.cluster_row_off.astype(np.int64))
    paths = np.diff(batch.cluster_path_off.astype(np.int64))
    ent_off = batch.grp_idx_off[batch.row_grp_off[batch.cluster_row_off.astype(np.int64)].astype(np.int64)].astype(np.int64)
    return (np.diff(ent_off) + rows * (paths + 1)).astype(np.float64)


def partition_clusters(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Greedy longest-processing-time bin packing: clusters by descending cost, each to the least loaded
    rank (ties: lower rank).  Deterministic; every rank computes the same partition."""
    order = sorted(range(len(costs)), key=lambda k: (-float(costs[k]), k))
    loads = [0.0] * world_size
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for k in order:
        r = min(range(world_size), key=lambda i: (loads[i], i))
        parts[r].append(k)
        loads[r] += float(costs[k])
    for p in parts:
        p.sort()
    return parts


def shard_batch(batch: ClusterBatch, rank: int, world_size: int):
    """(sub-batch of this rank, global cluster indices it holds)."""
    parts = partition_clusters(cluster_costs(batch), world_size)
    return batch.select(parts[rank]), parts[rank]


def all_gather_ragged(local: np.ndarray, dist, device: str = "cpu") -> List[np.ndarray]:
    """all_gather of 1-D float64 arrays whose lengths differ between ranks."""
    import torch
    world = dist.get_world_size()
    n_local = torch.tensor([local.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(max(sizes), 1)
    buf = torch.zeros(n_max, dtype=torch.float64, device=device)
    if local.size:
        buf[:local.size] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64)).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return [o[:n].cpu().numpy() for o, n in zip(out, sizes)]


def gather_cluster_values(local_values: Sequence[np.ndarray], local_clusters: Sequence[int], num_clusters: int, dist,
                          device: str = "cpu") -> List[np.ndarray]:
    """Every rank contributes one float64 vector per cluster it owns (e.g. the cluster's abundances);
    returns the vectors of all clusters in global cluster order, on every rank."""
    lens = np.array([len(v) for v in local_values], dtype=np.float64)
    flat = np.concatenate(local_values) if len(local_values) else np.zeros(0)
    ids = np.asarray(local_clusters, dtype=np.float64)
    g_ids = all_gather_ragged(ids, dist, device)
    g_lens = all_gather_ragged(lens, dist, device)
    g_flat = all_gather_ragged(flat, dist, device)
    out: List[np.ndarray] = [np.zeros(0)] * num_clusters
    for ids_r, lens_r, flat_r in zip(g_ids, g_lens, g_flat):
        off = 0
        for k, n in zip(ids_r.astype(np.int64), lens_r.astype(np.int64)):
            out[int(k)] = flat_r[off:off + int(n)].copy()
            off += int(n)
    return out


# ---- one giant cluster, rows spread over the ranks ---------------------------------------------------

def row_shard(num_rows: int, rank: int, world_size: int):
    """Contiguous row range [begin, end) of `rank` when the rows of ONE cluster are spread over the ranks
    ("one giant cluster"): every EM iteration then needs the all-reduce of C partial column
    sums that rpvg_hip_em_dense_sharded queues on its stream."""
    return (num_rows * rank) // world_size, (num_rows * (rank + 1)) // world_size


def broadcast_bytes(payload, nbytes: int, dist, device: str = "cpu", src: int = 0) -> bytes:
    """Hands `payload` (bytes on rank `src`, ignored elsewhere) to every rank."""
    import torch
    if dist.get_rank() == src:
        assert len(payload) == nbytes
        t = torch.tensor(list(payload), dtype=torch.uint8, device=device)
    else:
        t = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    return bytes(t.cpu().tolist())


def init_engine_comm(ctx, dist, device: str = "cpu"):
    """Creates the engine's own RCCL communicator on `ctx` (include/rpvg_hip.h, rpvg_hip_comm_*): rank 0 draws
    the id, torch.distributed carries it to the other ranks, every rank joins."""
    from . import hip
    uid = hip.Context.comm_unique_id() if dist.get_rank() == 0 else None
    uid = broadcast_bytes(uid, hip.COMM_ID_BYTES, dist, device)
    ctx.comm_init(uid, dist.get_world_size(), dist.get_rank())


# ---- the TPM denominator over sharded clusters ---------------------------------------------------------

def local_transcript_count(estimates, batch: ClusterBatch) -> float:
    """sum over this rank's clusters of abundance / effective length (src/main.cpp:1029-1057): the TPM denominator
    before it is summed over ranks.  estimates[k] belongs to cluster k of `batch`."""
    total = 0.0
    for k, e in enumerate(estimates):
        p0 = int(batch.cluster_path_off[k])
        lengths = batch.path_effective_length[p0:int(batch.cluster_path_off[k + 1])]
        members = [p for s in e.path_group_sets for p in s]
        if len(e.abundances) != len(members):  # one abundance per set (single-path sets) or none
            members = [s[0] for s in e.path_group_sets] if len(e.abundances) == len(e.path_group_sets) else []
        for ab, p in zip(e.abundances, members):
            if lengths[p] > 0:
                total += float(ab) / float(lengths[p])
    return total


def total_transcript_count(local_value: float, dist, device: str = "cpu") -> float:
    """The one scalar of the multi-GPU path that needs a collective besides the result gather: every rank needs
    the global TPM denominator to write its share of the output (all-reduce of one double; RCCL on GPUs)."""
    import torch
    t = torch.tensor([local_value], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return float(t.item())
