"""world_size-2 test of the multi-GPU path's host logic on CPU (gloo): deterministic cluster
partition, per-rank shards, ragged all_gather of per-cluster results back into global order.
The per-rank compute stand-in is the CPU oracle (the GPU engine needs a GPU); what is under test is
the sharding and the gather."""
import os
import socket

import numpy as np
import pytest

from rpvg_amd import dist as rdist, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    from rpvg_amd.batch import make_params
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch = synth.generate(seed=21, num_clusters=40, total_paths=1200, total_reads=30000)
        shard, mine = rdist.shard_batch(batch, rank, world)
        est, _ = pyoracle.run("transcripts", make_params(), shard, 1)
        gathered = rdist.gather_cluster_values([e.abundances for e in est], mine, batch.num_clusters, dist)
        noise = rdist.gather_cluster_values([np.array([e.noise_count, e.total_count]) for e in est], mine,
                                            batch.num_clusters, dist)
        if rank == 0:
            full, _ = pyoracle.run("transcripts", make_params(), batch, 2)
            ok = all(np.array_equal(g, f.abundances) for g, f in zip(gathered, full))
            ok = ok and all(n[0] == f.noise_count and n[1] == f.total_count for n, f in zip(noise, full))
            with open(os.path.join(out_dir, "result"), "w") as f:
                f.write("ok" if ok else "mismatch")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    batch = synth.generate(seed=22, num_clusters=200, total_paths=8000, total_reads=200000)
    costs = rdist.cluster_costs(batch)
    for world in (1, 2, 4, 8):
        parts = rdist.partition_clusters(costs, world)
        assert sorted(k for p in parts for k in p) == list(range(200))
        loads = [costs[p].sum() for p in parts]
        assert max(loads) <= min(loads) + costs.max() + 1e-9  # LPT guarantee
    assert rdist.partition_clusters(costs, 4) == rdist.partition_clusters(costs, 4)


def test_two_rank_gloo_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(str(tmp_path), "result")).read() == "ok"
