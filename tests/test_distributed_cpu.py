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
        tpm_den = rdist.total_transcript_count(rdist.local_transcript_count(est, shard), dist)
        if rank == 0:
            full, _ = pyoracle.run("transcripts", make_params(), batch, 2)
            want_den = rdist.local_transcript_count(full, batch)
            assert abs(tpm_den - want_den) <= 1e-9 * want_den and want_den > 0
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


def _row_shard_worker(rank, world, port, out_dir):
    """Model of the row-sharded EM protocol of rpvg_hip_em_dense_sharded (em_dense.hip): every rank streams its
    own rows, the C partial column sums are all-reduced, every rank applies the same update and convergence rule
    (src/path_abundance_estimator.cpp:58-113).  Checks that the protocol reproduces the unsharded oracle —
    abundances and the iteration it stops at — and exercises the helpers the GPU path uses (row_shard,
    broadcast_bytes) over a real two-process group."""
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        payload = bytes(range(128)) if rank == 0 else None
        got = rdist.broadcast_bytes(payload, 128, dist)
        assert got == bytes(range(128))

        rng = np.random.default_rng(5)  # same matrix on both ranks; each uses its shard only
        R, N = 3001, 37
        P = rng.random((R, N)) * (rng.random((R, N)) < 0.3)
        P[P.sum(axis=1) == 0, 0] = 0.5
        noise = rng.choice([1e-4, 1e-3, 0.1], size=R)
        P = P / P.sum(axis=1, keepdims=True) * (1 - noise)[:, None]
        P = np.concatenate([P, noise[:, None]], axis=1)
        counts = rng.integers(1, 5, size=R).astype(np.float64)
        total = counts.sum()
        r0, r1 = rdist.row_shard(R, rank, world)
        assert (r0, r1) == ((0, R // 2) if rank == 0 else (R // 2, R))
        Pl, cl = P[r0:r1], counts[r0:r1]

        C_ = N + 1

        def solve(poison_rank=None, fail_checks_rank=None):
            """The protocol with its status words: a handshake before the EM (sum over ranks of "my local checks failed") and
            one word behind the C column sums of every all-reduce (a rank whose sums are not finite raises it): no rank
            leaves a collective its peers are still going to enter."""
            hand = torch.tensor([1.0 if rank == fail_checks_rank else 0.0], dtype=torch.float64)
            dist.all_reduce(hand)
            if hand.item() != 0:
                return "peer failed its checks", None, 0
            a = np.full(C_, np.float64(np.float32(1.0) / np.float32(C_)))
            if rank == poison_rank:
                a[:] = np.nan
            conv_its, its = 0, 0
            while its < 10000:
                with np.errstate(all="ignore"):
                    s = Pl @ a
                    local = (cl / s) @ Pl
                t = torch.from_numpy(np.concatenate([local, [0.0 if np.all(np.isfinite(local)) else 1.0]]))
                dist.all_reduce(t)  # the one exchange step of the path: C sums + the status word
                if t[-1].item() != 0:
                    return "not finite", None, its
                an = a * t.numpy()[:-1] / total
                its += 1
                big = an >= 1e-8
                viol = bool(np.any(np.abs(an[big] - a[big]) / an[big] > 1e-3))
                a = an
                conv_its = 0 if viol else conv_its + 1
                if conv_its == 10:
                    break
            return "ok", a, its

        # a rank that fails its checks, a rank with poisoned values: every rank returns the same verdict, nobody hangs
        assert solve(fail_checks_rank=1)[0] == "peer failed its checks"
        verdict, _, stopped = solve(poison_rank=1)
        assert verdict == "not finite" and stopped == 0
        verdict, a, its = solve()
        assert verdict == "ok"
        ab = np.where(a[:-1] < 1e-8, 0.0, a[:-1] * total)

        # every rank must hold the same bits (same stop iteration is what keeps the collective calls matched)
        mine = torch.from_numpy(np.concatenate([a, [float(its)]]))
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert all(torch.equal(both[0], b) for b in both)

        if rank == 0:
            ref_ab, ref_noise, ref_total, ref_its, _ = pyoracle.em_dense(P, counts)
            ok = its == ref_its and np.allclose(ab, ref_ab, rtol=1e-9, atol=1e-12)
            ok = ok and abs(ab.sum() + (total - ab.sum()) - ref_total) < 1e-6
            with open(os.path.join(out_dir, "result_rows"), "w") as f:
                f.write("ok" if ok else f"mismatch its {its} vs {ref_its}")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_row_shard_covers_all_rows():
    for R in (1, 7, 1000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [rdist.row_shard(R, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == R
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_row_sharded_em_protocol(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_row_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(str(tmp_path), "result_rows")).read() == "ok"


def _run_bench(extra_args, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    # the collectives over gloo and the per-rank compute by the oracle: what is under test is bench.py's launcher,
    # its rank protocol (barrier, max over ranks, sums, gathers) and the line it prints
    env.update(RPVG_BENCH_DIST_BACKEND="gloo", RPVG_BENCH_ENGINE="tests.oracle_engine_stub", PYTHONPATH=root, OMP_NUM_THREADS="1")
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--scale", "0.01",
                          "--no-cpu-baseline"] + extra_args, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a torch.distributed.run environment starts two ranks itself."""
    one = _run_bench(["--gpus", "1"])
    two = _run_bench(["--gpus", "2"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "weak" and two["config"]["clusters_per_gpu"] == one["config"]["clusters_per_gpu"]
    # a stand-in engine claims no metric: the number the protocol computed is kept under another name
    assert two["mass_conserved"] and two["value"] is None and two["protocol_value"] > 0
    assert two["engine_module"] == "tests.oracle_engine_stub" and two["metric"].startswith("none")
    # weak scaling: every rank owns a batch of its own, the gathered mass is that of both
    assert two["gathered_abundance_mass"] > 1.5 * 0.9 * 10000000 * 0.01
    assert two["tpm_denominator"] > 0


def test_bench_strong_scaling_shards_one_batch():
    two = _run_bench(["--gpus", "2", "--scaling", "strong"])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["mass_conserved"]
    reads = 10000000 * 0.01  # ONE batch, its clusters sharded over the ranks: the gathered abundances are those of one batch
    assert 0.9 * reads < two["gathered_abundance_mass"] <= reads


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", RPVG_BENCH_DIST_BACKEND="gloo",
               RPVG_BENCH_ENGINE="tests.oracle_engine_stub", PYTHONPATH=root)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--scale", "0.01"], env=env,
                         capture_output=True, text=True, timeout=120, cwd=root)
    assert out.returncode != 0 and "--gpus 2" in (out.stderr + out.stdout)


def test_numa_binding_leaves_the_process_alone_when_it_cannot_be_read(monkeypatch):
    """bench.py binds a rank of a multi-rank run to the CPUs of its GPU's NUMA node; a GPU whose node cannot be read (no such
    PCI device here), or a node with too few of the process's CPUs, changes nothing."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class Props:
        pci_domain_id, pci_bus_id, pci_device_id = 0xfffe, 0xfe, 0x1e

    class Cuda:
        @staticmethod
        def get_device_properties(index):
            return Props()

    class Torch:
        cuda = Cuda()

    before = os.sched_getaffinity(0)
    bench.bind_to_the_gpus_numa_node(Torch(), 0)
    assert os.sched_getaffinity(0) == before and bench.NUMA_BINDING is None
