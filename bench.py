#!/usr/bin/env python3
"""bench.py — read-pairs quantified per second on the synthetic pantranscriptome.

A "step" is one pass of the inference hot path over one batch of path clusters:
NestedPathAbundanceEstimator::estimateBatch() (`-i haplotype-transcripts`, reference defaults) on the "10M read pairs
x 200k paths in ~5k clusters" workload (BASELINE.json configs[2]).  `value` is SURVEY.md section 8(d)'s metric: read pairs
per second of wall clock with the H2D of every batch's sparse rows and the D2H of its results inside the clock (steady
state of a double-buffered pipeline: batch n + 1 is uploaded under the kernels of batch n; the first batch is resident
when the clock starts).  The same K steps on a batch that stays resident are reported next to it (`value_resident`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s3|c2] [--model ...] [--scale F]

One process per GPU: started under torch.distributed.run (WORLD_SIZE in the environment) this process is one
rank; started plainly with --gpus N > 1 it launches the N ranks itself (torch.distributed.run, rendezvous on
127.0.0.1) and passes their output through.  Clusters are independent, so there is no data-path collective:
every rank owns a full-size batch of its own (weak scaling) and the final per-path abundance vectors are
gathered over RCCL after the timed region.  Rank 0 prints ONE JSON line; its n_gpus is the number of ranks
that ran and must equal --gpus.

--workload c2 runs the other bench-able configuration, "1M read pairs x 2k paths, one dense cluster"
(configs[1], 16 GB FP64 streamed from HBM every EM iteration, fixed 50-iteration budget).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rpvg_amd  # noqa: E402,F401  before torch touches the GPU: runtime settings (hardware queues, heap top pad)

# OpenMP teams (host layer, CPU oracle) sleep instead of spinning between parallel regions, so that
# idle workers do not compete with the thread that drives the GPU.  Must be set before libgomp loads.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy kernel reaches
FP64_VALU_PEAK_TFLOPS = 78.6  # FP64 vector rate = half the 157 TF FP32 vector rate (MI355X_MICROARCH.md)


PROFILE_ROUND = "r06"


def load_pmc(name):
    """A PMC summary of profiles/<round>/ (separate rocprofv3 --pmc passes, tools/refresh_profiles_r03.sh): static
    records of the tree of the refresh — the commit they were taken on is part of the record."""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, name)
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "r05", name)  # (the round before: the record names its commit)
    if not os.path.exists(path):
        return None
    try:
        record = json.load(open(path))
    except (OSError, ValueError):
        return None
    if isinstance(record, dict):  # (where the record lies, whatever it says of itself)
        made_by = str(record.get("source", "rocprofv3 --pmc passes"))
        if "(" in made_by and made_by.endswith(")"):
            made_by = made_by[made_by.rindex("(") + 1:-1]
        record["source"] = os.path.relpath(path, ROOT) + " (" + made_by + ")"
    return record


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="s3", choices=["s3", "c2", "s5", "rows", "e2e", "a1"],
                    help="s3: BASELINE.json configs[2]/[3] (default, the metric's configuration); c2: configs[1] single dense "
                         "cluster; s5: configs[4] diploid haplotype Gibbs, 10M reads x 500k paths; rows: the step before the path "
                         "(alignment paths -> merged rows, SURVEY.md 8f rank 2) on the configs[2] reads; e2e: rows + estimates in one "
                         "pass, nothing leaving the GPU in between")
    ap.add_argument("--model", default="haplotype-transcripts", choices=["haplotype-transcripts", "transcripts", "haplotypes"])
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the full workload (parity/dev runs only)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns a full-size batch; strong: ONE batch, clusters sharded over the ranks")
    ap.add_argument("--in-flight", type=int, default=1, choices=[1, 2, 3, 4],
                    help="s3/s5: batches in flight per GPU.  1 = a step returns before the next one starts (default); 2 = two "
                         "engines on the GPU, each on its own resident copy of the batch, driven by two host threads: the K steps "
                         "overlap (one batch's host prologue and epilogue run under the other's kernels)")
    ap.add_argument("--pipeline-workers", type=int, default=0,
                    help="s3/s5 headline: estimator threads of the batch pipeline (rpvg_amd/host/batch_pipeline.hpp) = batches estimated side by "
                         "side; 0 = the library's default (4)")
    ap.add_argument("--team", type=int, default=1024,
                    help="a1: threads of the OpenMP team that calls PathEstimator::estimate() (the reference's -t; the threads sleep in the "
                         "call combiner while their batch is on the GPU, so a team far larger than the cores is what keeps batches in flight)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


# Test seams (tests/test_distributed_cpu.py drives the launcher and the rank protocol without a GPU): the collective
# backend and the module that provides Engine.  The defaults are the product: RCCL and the HIP engine, which fails
# loudly without a GPU.  Nothing in this file imports the oracle except the cpu_baseline leg.
DIST_BACKEND = os.environ.get("RPVG_BENCH_DIST_BACKEND", "nccl")
ENGINE_MODULE = os.environ.get("RPVG_BENCH_ENGINE", "rpvg_amd.engine")
DEVICE = "cuda" if DIST_BACKEND == "nccl" else "cpu"


def launch_ranks(n_gpus):
    """--gpus N > 1 without a torch.distributed.run environment: start the N ranks of this node and wait for them."""
    import socket
    import subprocess
    with socket.socket() as sock:  # a free port for the rendezvous
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the ranks of this node
    return subprocess.call(cmd, env=env)


NUMA_BINDING = None  # what bind_to_the_gpus_numa_node() did, for the line


def bind_to_the_gpus_numa_node(torch, local_rank):
    """Several ranks on a node: the threads of a rank (its uploader copies 190 MB per batch out of host memory, 40-50 GB/s per rank)
    on the CPUs of the NUMA node its GPU hangs on, before anything is allocated (first touch puts the batch there too).  Nothing
    happens when the node cannot be read, has fewer than six of the CPUs the process may use, or RPVG_BENCH_NO_NUMA_BIND is set."""
    global NUMA_BINDING
    if os.environ.get("RPVG_BENCH_NO_NUMA_BIND") or not hasattr(os, "sched_setaffinity"):
        return
    try:
        props = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        mine = set(os.sched_getaffinity(0)) & cpus
        if len(mine) < 6:
            return
        os.sched_setaffinity(0, mine)
        NUMA_BINDING = dict(gpu=bdf, numa_node=node, cpus=len(mine))
    except (OSError, ValueError, AttributeError, RuntimeError):
        return


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    import torch
    if world > 1 and DEVICE == "cuda" and torch.cuda.is_available():
        bind_to_the_gpus_numa_node(torch, local_rank)
    dist = None
    if world > 1 or os.environ.get("RPVG_BENCH_FORCE_DIST"):  # the env var exercises the RCCL path with one rank
        import torch.distributed as dist_mod
        if DEVICE == "cuda":
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group(backend=DIST_BACKEND, device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend=DIST_BACKEND)
        dist = dist_mod
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world, dist, torch


def barrier_sync(dist, torch):
    if dist is not None:
        dist.barrier()
    if DEVICE == "cuda":
        torch.cuda.synchronize()


def max_over_ranks(x, dist, torch):
    if dist is None:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=DEVICE)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, dist, torch):
    if dist is None:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=DEVICE)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def cpu_quota():
    """CPUs' worth of time the container may use if a cgroup limits it (cpu.max), else None."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    return None


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs of /proc/cpuinfo); hardware threads if that fails."""
    try:
        pairs, package = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                package = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((package, line.split(":")[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_quota_note():
    quota = cpu_quota()
    return f"; cgroup cpu.max grants {quota:.0f} CPUs' worth of time" if quota else ""


def cpu_baseline_s3(batch, model, params, target_seconds):
    """The CPU oracle (OpenMP over clusters, serial inside a cluster, as src/main.cpp:829) on a strided sample of the
    same batch, sized for about target_seconds of CPU work.  Threads = the cores this process can actually use:
    min(physical cores, cgroup quota); the line with every hardware thread (oversubscribed when a quota applies) is
    kept next to it as `all_threads`."""
    from oracle import pyoracle
    hardware_threads = os.cpu_count() or 1
    quota = cpu_quota()
    cores = max(1, min(physical_cores(), int(quota) if quota else hardware_threads))
    K = batch.num_clusters

    def timed(threads):
        probe_idx = list(range(0, K, max(1, K // 40)))
        _, probe_secs = pyoracle.run(model, params, batch.select(probe_idx), threads)
        est_full = probe_secs * K / max(1, len(probe_idx))
        stride = max(1, int(round(est_full / target_seconds)))
        idx = list(range(0, K, stride))
        sample = batch.select(idx)
        _, secs = pyoracle.run(model, params, sample, threads)
        return dict(value=sample.total_reads / secs, unit="read-pairs/s", cores=threads, threads=threads, kind="port",
                    sample=f"every {stride}th cluster of the batch ({len(idx)} clusters, {sample.total_reads} read pairs, {secs:.2f} s): "
                           f"oracle/ C++ restatement, OpenMP dynamic over clusters, {threads} threads")

    line = timed(cores)
    line.update(physical_cores=physical_cores(), hardware_threads=hardware_threads, quota_cpus=quota)
    if hardware_threads != cores:
        every = timed(hardware_threads)
        line["all_threads"] = dict(value=every["value"], threads=hardware_threads, sample=every["sample"])
    return line


def merge_stats(into, other):
    for key, value in other.items():
        if isinstance(value, dict):
            merge_stats(into.setdefault(key, {}), value)
        else:
            into[key] = into.get(key, 0) + value


def host_threads_per_lane():
    """The OpenMP team of one host lane (rpvg_amd/host/pipeline_lanes.hpp, hostThreads()), as the library computes it."""
    try:
        from rpvg_amd import engine
        return int(engine.lib().rpvg_amd_host_threads())
    except Exception:  # noqa: BLE001  (stand-in engine of the launcher test)
        return None


def run_s3(args, rank, local_rank, world, dist, torch):
    import numpy as np
    import importlib
    from rpvg_amd import dist as rdist, synth
    from rpvg_amd.batch import make_params
    eng_mod = importlib.import_module(ENGINE_MODULE)

    s5 = args.workload == "s5"
    if s5:
        # BASELINE.json configs[4] / SURVEY.md §8d S5: -i haplotypes -y 2 --use-hap-gibbs
        args.model = "haplotypes"
        params = make_params(use_hap_gibbs=1)
        base_paths, seed0, gen_kw, config_name = 500000, 5, dict(max_cluster_paths=5000), "configs[4]"
    else:
        params = make_params()
        if os.environ.get("RPVG_BENCH_MAX_EM_ITS"):  # (experiment: what the EM's iterations cost the step; not a measurement of record)
            params.max_em_its = int(os.environ["RPVG_BENCH_MAX_EM_ITS"])
        base_paths, seed0, gen_kw, config_name = 200000, 3, {}, "configs[2]"
    K = max(8, int(round(5000 * args.scale)))
    total_paths = max(K, int(round(base_paths * args.scale)))
    total_reads = int(round(10000000 * args.scale))
    if args.scaling == "weak" or world == 1:
        # weak scaling: every rank owns a full-size batch (its own seed)
        batch = synth.generate(seed=seed0 + rank, num_clusters=K, total_paths=total_paths, total_reads=total_reads, **gen_kw)
        my_clusters = list(range(batch.num_clusters))
        global_clusters = batch.num_clusters
    else:
        # strong scaling (BASELINE.json configs[3]): the same batch on every rank, clusters bin-packed over ranks
        full = synth.generate(seed=seed0, num_clusters=K, total_paths=total_paths, total_reads=total_reads, **gen_kw)
        batch, my_clusters = rdist.shard_batch(full, rank, world)
        global_clusters = full.num_clusters

    eng = eng_mod.Engine(local_rank)
    t_up = time.perf_counter()
    prepared = eng.prepare(batch)  # upload: inputs are resident in HBM before the timed region
    upload_ms = (time.perf_counter() - t_up) * 1e3
    # engine start-up (second host lane and its device context, scratch pools): one pass outside the measurement,
    # so that the W warmup steps are warmup and not initialisation
    eng.run_raw(args.model, params, prepared)
    for _ in range(args.warmup):
        eng.run_raw(args.model, params, prepared)
    others = []  # --in-flight 2: a second engine with its own resident copy of the batch
    for _ in range(args.in_flight - 1):
        other = eng_mod.Engine(local_rank)
        other_prepared = other.prepare(batch)
        for _ in range(args.warmup + 1):
            other.run_raw(args.model, params, other_prepared)
        other.reset_stats()
        others.append((other, other_prepared))
    eng.reset_stats()

    barrier_sync(dist, torch)
    t0 = time.perf_counter()
    step_ms = []
    if not others:
        for _ in range(args.steps):
            step_ms.append(eng.run_raw(args.model, params, prepared) * 1e3)
    else:
        import threading
        lanes = [(eng, prepared)] + others
        share = [args.steps // len(lanes) + (1 if k < args.steps % len(lanes) else 0) for k in range(len(lanes))]
        errors = []

        def drive(k):
            try:
                for _ in range(share[k]):
                    lanes[k][0].run_raw(args.model, params, lanes[k][1])  # ctypes releases the GIL during the call
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        threads = [threading.Thread(target=drive, args=(k,)) for k in range(len(lanes))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
    barrier_sync(dist, torch)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, torch)
    stats = eng.stats()
    for other, _ in others:
        merge_stats(stats, other.stats())
    if os.environ.get("RPVG_BENCH_STEP_TIMES"):  # spread of the single steps (the JSON line reports the mean)
        print("step ms:", " ".join(f"{t:.1f}" for t in step_ms), file=sys.stderr)

    reads_all = sum_over_ranks(float(batch.total_reads), dist, torch)

    # The headline: the same K steps with the rows of every batch arriving from host memory (SURVEY.md §8d counts the H2D
    # of the sparse rows in): page-locked host arrays, two resident slots, an uploader engine with a stream of its own —
    # batch n + 1 is validated, copied and expanded under the kernels of batch n.  The kernel statistics of the line are
    # those of this loop.
    h2d = None
    if not others and DEVICE == "cuda" and hasattr(prepared, "reupload"):
        quick = bool(os.environ.get("RPVG_BENCH_NO_SINGLE")) and hasattr(eng_mod, "Pipeline")  # (the pipeline only: profiling runs, the s5 line inside the default run)
        single = None if quick else measure_with_uploads(args, eng, eng_mod, prepared, batch, params, local_rank, dist, torch)
        single_stats = single.pop("stats") if single else {}
        if quick:
            h2d = measure_pipeline(args, eng_mod, batch, params, local_rank, dist, torch)
            stats = h2d.pop("stats")
        elif hasattr(eng_mod, "Pipeline") and not os.environ.get("RPVG_BENCH_NO_PIPELINE"):
            # the headline: the same K batches through the pipeline of the host library (rpvg_amd/host/batch_pipeline.hpp) — an uploader
            # thread and several single-lane engines, several batches in flight on the GPU, every batch uploaded inside the clock
            h2d = measure_pipeline(args, eng_mod, batch, params, local_rank, dist, torch)
            stats = h2d.pop("stats")
            one = {k: single[k] for k in ("ms_per_step_with_h2d", "ms_per_step_with_h2d_serial", "ms_per_step_spread", "host_cpu_ms_per_step",
                                          "h2d_ms_per_batch", "h2d_gb_per_s")}
            if single_stats.get("busy_ms") is not None:
                one["gpu_active_frac"] = (single_stats["busy_ms"] / args.steps) / single["ms_per_step_with_h2d"]
            one["note"] = ("the same steps with ONE batch in flight (the headline of rounds 2-4): estimateBatch on two host lanes, the next batch "
                           "uploaded under it from a second resident slot")
            h2d["one_batch_in_flight"] = one
            # (the configs[2] line only: a model that draws random numbers gives cluster i of a SUBMITTED batch the generator mt19937(rng_seed + i),
            # src/main.cpp:976 — the parts of a data set then draw from other generators than the whole)
            if world == 1 and args.workload == "s3" and args.model == "haplotype-transcripts" and not os.environ.get("RPVG_BENCH_NO_SINGLE_DATASET"):
                try:
                    h2d["single_dataset"] = measure_single_dataset(args, eng_mod, batch, params, local_rank)
                except Exception as exc:  # noqa: BLE001
                    h2d["single_dataset"] = dict(error=str(exc))
            if world == 1 and args.scale >= 1.0 and not os.environ.get("RPVG_BENCH_NO_HOST_BOUND"):
                try:
                    bound = host_bound_line(args, eng_mod, batch, params, local_rank)
                    bound["efficiency_bound"] = min(1.0, h2d["ms_per_step_with_h2d"] / bound["ms_per_step"])
                    h2d["predicted_host_bound_8_ranks"] = bound
                except Exception as exc:  # noqa: BLE001
                    h2d["predicted_host_bound_8_ranks"] = dict(error=str(exc))
        else:
            h2d, stats = single, single_stats

    # after the timed region: gather the per-path abundances of every rank over RCCL (what a multi-GPU
    # driver does before writing rpvg.txt); also fetch one decoded result for a sanity check
    est, _ = eng.run(args.model, params, prepared)
    pipeline_est = h2d.pop("sample_estimates", None) if h2d is not None else None
    single_dataset = h2d.get("single_dataset") if h2d is not None else None
    if single_dataset and "estimates" in single_dataset:  # the parts' estimates are the whole batch's, bit for bit
        parts_est = single_dataset.pop("estimates")
        single_dataset["equal_whole_batch"] = bool(len(parts_est) == len(est) and all(
            a.path_group_sets == b.path_group_sets and np.array_equal(a.posteriors, b.posteriors) and np.array_equal(a.abundances, b.abundances)
            and a.noise_count == b.noise_count and a.em_iters == b.em_iters for a, b in zip(parts_est, est)))
    if pipeline_est is not None:  # the pipeline's batches are the engine's: same estimates, to the bit
        assert len(pipeline_est) == len(est)
        def same(a, b):
            return (a.path_group_sets == b.path_group_sets and np.allclose(a.posteriors, b.posteriors, rtol=1e-9, atol=1e-12)
                    and np.allclose(a.abundances, b.abundances, rtol=1e-9, atol=1e-9) and abs(a.noise_count - b.noise_count) <= 1e-9 * max(1.0, b.total_count)
                    and a.em_iters == b.em_iters)
        pipeline_equal = all(same(a, b) for a, b in zip(pipeline_est, est))
        if not pipeline_equal:
            bad = [k for k, (a, b) in enumerate(zip(pipeline_est, est)) if not same(a, b)]
            print(f"pipeline != single engine in {len(bad)} clusters, first {bad[:5]}", file=sys.stderr)
    else:
        pipeline_equal = None
    if args.model == "haplotypes":  # posteriors only: they sum to one per cluster
        mass_ok = all(abs(e.posteriors.sum() - 1) <= 1e-6 for e in est if e.total_count > 0 and len(e.posteriors))
    else:
        mass_ok = all(abs(e.abundances.sum() + e.noise_count - e.total_count) <= 1e-6 * max(1.0, e.total_count) for e in est)
    gathered = None
    tpm_denominator = None
    if dist is not None and args.model != "haplotypes":
        # the other collective of a multi-GPU run: the TPM denominator (src/main.cpp:1029-1057) summed over ranks
        tpm_denominator = rdist.total_transcript_count(rdist.local_transcript_count(est, batch), dist, DEVICE)
    if dist is not None:
        if args.scaling == "strong":
            per_cluster = rdist.gather_cluster_values([e.abundances for e in est], my_clusters, global_clusters, dist, DEVICE)
            gathered = float(sum(float(v.sum()) for v in per_cluster))
        else:
            flat = np.concatenate([e.abundances for e in est]) if est else np.zeros(0)
            gathered = float(sum(float(v.sum()) for v in rdist.all_gather_ragged(flat, dist, DEVICE)))

    if rank != 0:
        return None

    ms_resident = elapsed / args.steps * 1e3
    ms_per_step = h2d["ms_per_step_with_h2d"] if h2d is not None else ms_resident
    value = reads_all / (ms_per_step / 1e3)
    # the dominant EM kernel: the variant with the most device time (its own HIP events on its own stream)
    em_kernels = {name: ks for name, ks in stats.get("em_kernel", {}).items() if ks["launches"]}
    traffic = None
    pmc = load_pmc("pmc_traffic_s3.json") if (args.scale == 1.0 and args.workload == "s3" and args.model == "haplotype-transcripts") else None
    if em_kernels:
        dominant = max(em_kernels, key=lambda name: em_kernels[name]["ms"])
        dk = em_kernels[dominant]
        em_ms = dk["ms"] / dk["launches"]
        em_bytes = dk["alg_bytes"] / dk["launches"]
    else:  # a stand-in engine without per-kernel statistics
        dominant = "emSparseKernel"
        em_ms = stats["em_sparse_ms"] / max(1, stats["em_sparse_launches"])
        em_bytes = stats["em_sparse_alg_bytes"] / max(1, stats["em_sparse_launches"])
    achieved = (em_bytes / 1e9) / (em_ms / 1e3) if em_ms > 0 else 0.0
    if pmc is not None:
        # separate rocprofv3 --pmc passes of this command (tools/pmc_traffic.py), per kernel variant and launch like `achieved`
        per_kernel = pmc.get("per_kernel", {})
        for name, rec in per_kernel.items():
            if name.replace(" ", "").startswith(dominant.replace(" ", "")):
                traffic = rec["traffic_bytes_per_launch"]
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, kernel=dominant, ms_per_launch=em_ms, algorithmic_bytes_per_launch=em_bytes,
                    note="dominant EM kernel = the launch with the most device time (emRegisterKernel: the five register-resident size "
                         "bins in one launch); ms_per_launch = its own HIP-event span on the "
                         "stream it runs on; algorithmic bytes = sum over its problems of iterations x (12 B/entry + 20 B/row + "
                         "16 B/column); the problems are register/LDS/L2-resident across iterations, so this is effective bandwidth "
                         "(the HBM traffic of the PMC passes is a fraction of the algorithmic bytes) and the kernel's real bound is the "
                         "latency of one EM iteration times the iteration count of its slowest problem (em_kernels.*.us_per_iteration_of_slowest)")
    if pmc is not None:
        roofline.update(pmc_commit=pmc.get("commit"), pmc_source=pmc.get("source"))
    em_kernel_lines = {}
    for name, ks in em_kernels.items():
        n = ks["launches"]
        em_kernel_lines[name] = dict(
            ms_per_launch=ks["ms"] / n, launches_per_step=n / args.steps, problems_per_launch=ks["problems"] / n,
            iterations_per_launch=ks["iterations"] / n, slowest_problem_iterations=ks["max_iterations"] / n,
            us_per_iteration_of_slowest=(ks["ms"] * 1e3 / ks["max_iterations"]) if ks["max_iterations"] else None,
            algorithmic_bytes_per_launch=ks["alg_bytes"] / n,
            effective_gb_per_s=(ks["alg_bytes"] / 1e9) / (ks["ms"] / 1e3) if ks["ms"] > 0 else 0.0)
    kernels = dict(
        em_sparse_ms_per_step=stats["em_sparse_ms"] / args.steps, loglik_ms_per_step=stats["loglik_ms"] / args.steps,
        build_ms_per_step=stats["build_ms"] / args.steps, h2d_ms_per_step=stats["h2d_ms"] / args.steps,
        collapse_ms_per_step=stats.get("collapse_ms", 0.0) / args.steps,
        em_iterations_per_step=stats["em_iterations_total"] / args.steps,
        loglik_evals_per_step=stats["loglik_evals"] / args.steps,
        loglik_gevals_per_s=(stats["loglik_evals"] / 1e9) / (stats["loglik_ms"] / 1e3) if stats["loglik_ms"] > 0 else 0.0)
    line = dict(
        metric="read-pairs quantified/sec", value=value, unit="read-pairs/s", n_gpus=world, steps=args.steps,
        warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
        scaling=("strong" if (args.scaling == "strong" and world > 1) else "weak"), vs_baseline=None,
        dtype="f64", data="synthetic",
        config=dict(workload=f"synthetic pantranscriptome: {total_reads} read pairs x {total_paths} paths in {K} clusters per GPU "
                             f"(BASELINE.json {config_name}), -i {args.model}"
                             + (" -y 2 --use-hap-gibbs" if s5 else "") + ", reference defaults",
                    clusters_per_gpu=K, rows_per_gpu=batch.num_rows, entries_per_gpu=int(len(batch.path_idx)),
                    parallelism=f"clusters sharded, {world} rank(s), final abundance gather over RCCL",
                    batches_in_flight=(h2d or {}).get("pipeline", {}).get("workers", args.in_flight)),
        roofline=roofline, kernels=kernels, em_kernels=em_kernel_lines, mass_conserved=bool(mass_ok),
        upload_ms=upload_ms, ms_per_step_resident=ms_resident, value_resident=reads_all / (ms_resident / 1e3),
        engine_module=ENGINE_MODULE, host_threads_per_lane=host_threads_per_lane(),
        value_note="value = read pairs / steady-state step with the H2D of every batch's rows and the D2H of its results inside the "
                   "clock (SURVEY.md section 8d); value_resident = the same K steps on a batch that stays in HBM")
    if stats.get("busy_ms") is not None:
        # union of the timed spans of both host lanes' contexts over the wall time of the headline loop: an upper bound of
        # the share of the step during which the GPU had work of this engine (a span runs from the first command of a stage
        # to its last on the stage's stream); the rocprofv3 kernel trace under profiles/ gives the exact figure
        line["gpu_active_frac"] = (stats["busy_ms"] / args.steps) / ms_per_step
    if h2d is not None:
        line.update(h2d)
        second = line.get("instrumented_pass")
        if second and second.get("gpu_active_frac") is not None:  # (the busy union over the wall time of the pass it was taken in)
            line["gpu_active_frac"] = second["gpu_active_frac"]
    if pipeline_equal is not None:
        line["pipeline_estimates_equal_single_engine"] = bool(pipeline_equal)
    if line.get("single_dataset") and "ms" in line["single_dataset"]:
        line["single_dataset_ms"] = line["single_dataset"]["ms"]
    if ENGINE_MODULE != "rpvg_amd.engine":
        # a stand-in engine (the CPU test of the launcher and the rank protocol): no metric is claimed
        line.update(metric="none: stand-in engine %s, launcher/protocol self-test only" % ENGINE_MODULE, protocol_value=value, value=None,
                    value_resident=None)
    if gathered is not None:
        line["gathered_abundance_mass"] = gathered
    if tpm_denominator is not None:
        line["tpm_denominator"] = tpm_denominator
    # The log-likelihood kernels (diploid search, Gibbs conditionals) are FP64-issue bound, not HBM bound: the search reads
    # its matrices once and works from LDS.  One evaluation = one row of one column set: an add and a multiply on the
    # running-product path (2 flop, the figure used here), a ~22-instruction logarithm on the rest.
    evals = stats["loglik_evals"]
    ll_ms = stats["loglik_ms"]
    tflops = (2.0 * evals / 1e12) / (ll_ms / 1e3) if ll_ms > 0 else 0.0
    search = dict(bound="fp64_valu", achieved=tflops, peak=FP64_VALU_PEAK_TFLOPS, unit="TFLOP/s", frac=tflops / FP64_VALU_PEAK_TFLOPS,
                  traffic=None, evals_per_step=evals / args.steps, flops_per_eval=2, gevals_per_s=(evals / 1e9) / (ll_ms / 1e3) if ll_ms > 0 else 0.0,
                  ms_per_step=ll_ms / args.steps,
                  note="device time = HIP-event spans of the log-likelihood kernels on their streams (the spans of the two host lanes "
                       "overlap each other and the EM kernels); VALU utilisation and instructions per evaluation: profiles/ PMC passes")
    if stats.get("search_tile_launches"):
        # the kernel the evaluations belong to, by its own HIP-event spans (the instrumented pass): what `frac` is computed from —
        # 2 flop x evaluations per launch / its mean duration; the family's spans above also hold the kernels behind it and overlap
        # between the batches in flight
        tile_ms = stats["search_tile_ms"] / stats["search_tile_launches"]
        evals_per_launch = (evals / args.steps) / max(1e-9, stats["search_tile_launches"] / args.steps)
        tile_tflops = (2.0 * evals_per_launch / 1e12) / (tile_ms / 1e3) if tile_ms > 0 else 0.0
        search.update(frac_over_family_spans=search["frac"], achieved_over_family_spans=search["achieved"], achieved=tile_tflops,
                      frac=tile_tflops / FP64_VALU_PEAK_TFLOPS, kernel_ms_per_launch=tile_ms, kernel_launches_per_step=stats["search_tile_launches"] / args.steps,
                      frac_note="achieved / frac = 2 flop x row-pair evaluations of a launch / the mean duration of pairTile2Kernel by its own HIP events "
                                "(rpvg_hip_kernel_stats::search_tile_ms); *_over_family_spans = the same evaluations over the spans of the whole "
                                "log-likelihood family (tile kernel, the collapse's held-back stage, resolveTableKernel), which overlap between batches in flight")
    if stats.get("search_pairs_possible"):
        search.update(pairs_possible_per_step=stats["search_pairs_possible"] / args.steps, pairs_kept_per_step=stats["search_pairs_kept"] / args.steps,
                      pairs_evaluated_exhaustively_per_step=stats["search_pairs_table"] / args.steps,
                      pairs_note="possible = G(G+1)/2 per searched matrix; evaluated exhaustively = pairs of the matrices whose every pair is "
                                 "evaluated (pairTile2Kernel: all matrices up to 1024 columns) where the reference skips the first columns its "
                                 "bound prunes (the sequential search reaches 4.1 of these 4.4 G row-pair evaluations anyway); kept = pairs "
                                 "that survive the threshold")
    pmc_search = load_pmc("pmc_search_s3.json") if (not s5 and args.scale == 1.0 and args.model == "haplotype-transcripts") else None
    if pmc_search is not None:
        # separate rocprofv3 --pmc passes of this workload with one host lane (tools/pmc_search_summary.py)
        search.update(valu_instructions_per_eval=pmc_search["valu_instructions_per_eval"], valu_busy=pmc_search["valu_busy"],
                      pmc_source=pmc_search["source"], pmc_commit=pmc_search.get("commit"))
    if s5:
        # no EM on this path: the chains of the sampler run on the device (rpvg_hip_group_gibbs, round 3: mt19937 and libstdc++'s
        # distributions restated, draw for draw); the FP64 work is the log-likelihood contraction of the conditionals they ask for
        # two kinds of device time in the sampler: the conditionals (gibbsConditionalTileKernel in the first round, gibbsConditionalKernel
        # behind it: FP64 log-likelihood contractions, their own spans = loglik_ms) and everything else between the sampler's first
        # and last kernel (gibbsAdvanceKernel: the chains' draws, request offsets, distributions, the host's looks at the progress word:
        # gibbs_ms - loglik_ms).  The roofline object is that of the conditionals — the kernels with a flop count — and says which
        # of the two holds more of the sampler's span.
        chains_ms = max(0.0, stats.get("gibbs_ms", 0.0) - stats["loglik_ms"])
        search["kernel"] = "gibbsConditionalTileKernel + gibbsConditionalKernel"
        search["sampler_span_ms_per_step"] = stats.get("gibbs_ms", 0.0) / args.steps
        search["chains_and_bookkeeping_ms_per_step"] = chains_ms / args.steps
        search["dominant_by_device_time"] = ("conditionals (gibbsConditionalTileKernel + gibbsConditionalKernel)" if stats["loglik_ms"] >= chains_ms
                                             else "chains and bookkeeping (gibbsAdvanceKernel, gibbsRequestOffsetsKernel, gibbsDistributionKernel): latency bound, no flop count")
        search["note"] = ("device time = HIP-event spans of the conditional kernels (one launch per round of the sampler) on the estimator threads' streams, "
                          "summed over the batches in flight (they overlap); sampler_span = first to last kernel of rpvg_hip_group_gibbs")
        line["roofline"] = search
        line["sampler"] = "device: rpvg_hip_group_gibbs (RPVG_AMD_HOST_GIBBS=1: host-driven lock-step sampler, round 2)"
        line["ms_per_step_outside_conditionals"] = ms_per_step - (stats["loglik_ms"] + stats["build_ms"] + stats["h2d_ms"]) / args.steps
    else:
        search["kernel"] = "pairTile2Kernel + resolveTableKernel"
        line["roofline_search"] = search
    # the engines of this process go before the lines that other processes measure (configs[4], the drop-in path): their contexts hold
    # streams, hardware queues and resident batches on the GPU the children are timed on
    try:
        prepared.free()
        for other, other_prepared in others:
            other_prepared.free()
            other.close()
        eng.close()
    except Exception:  # noqa: BLE001  (a stand-in engine without these)
        pass
    if args.scale >= 1.0 and not s5 and DEVICE == "cuda":
        try:
            line["roofline_dense_em"] = dense_em_roofline(local_rank)
        except Exception as exc:  # the record is optional; the default workload's line stands on its own
            line["roofline_dense_em"] = dict(error=str(exc))
        if args.workload == "s3" and args.model == "haplotype-transcripts" and world == 1 and not os.environ.get("RPVG_BENCH_NO_GIBBS_LINE"):
            try:
                line["roofline_gibbs"] = gibbs_line(args, local_rank)
            except Exception as exc:  # noqa: BLE001
                line["roofline_gibbs"] = dict(error=str(exc))
        if args.workload == "s3" and args.model == "haplotype-transcripts" and world == 1 and not os.environ.get("RPVG_BENCH_NO_DROP_IN_LINE"):
            try:
                line["drop_in"] = drop_in_line(args, local_rank)
            except Exception as exc:  # noqa: BLE001
                line["drop_in"] = dict(error=str(exc))
            # ... and with a team of 256 (the reference's -t): the callers are blocked on the GPU, not computing, so the calls in flight —
            # and with them the clusters per batch — are the team's size, not the host's cores (tools/r06_a1_teams.sh: 64 ... 512)
            try:
                line["drop_in_team_256"] = drop_in_line(args, local_rank, team=256)
            except Exception as exc:  # noqa: BLE001
                line["drop_in_team_256"] = dict(error=str(exc))
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_s3(batch, args.model, params, args.cpu_seconds)
        cpu_value = line["cpu_baseline"].get("value") if isinstance(line["cpu_baseline"], dict) else None
        if cpu_value:
            # (vs_baseline stays null: BASELINE.md holds no published number for this metric — the ratio to the same-run CPU line is its own field)
            line["vs_cpu_baseline"] = line["value"] / cpu_value if line.get("value") else None
            for key in ("drop_in", "drop_in_team_256"):
                if isinstance(line.get(key), dict) and line[key].get("value"):
                    line[key]["vs_cpu_baseline"] = line[key]["value"] / cpu_value
    return line


def drop_in_line(args, local_rank, steps=5, team=64):
    """The reference's own call pattern inside the default run, so that the driver's record carries it: PathEstimator::estimate() once
    per cluster from an OpenMP team of `team` threads on the configs[2] workload (src/main.cpp:829,976-977) — a short run of the
    --workload a1 bench as a child process (its own engine, its own memory)."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "a1", "--team", str(team), "--steps", str(steps), "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-400:])
    d = json.loads(out.stdout.strip().splitlines()[-1])
    return dict(value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], ms_per_step_in_order=d["ms_per_step_in_order"], steps=d["steps"],
                team_threads=team, estimates_equal_estimate_batch=d["estimates_equal_estimate_batch"], workload=d["config"]["workload"],
                note="a step = all 5 000 clusters of the configs[2] data set through estimate(), one call per cluster, every caller blocked until its "
                     "cluster is done; callers flatten their clusters into page-locked segments that the GPU reads in place, the calls in flight are "
                     "joined into batches (PathEstimator::CallCombiner); python bench.py --workload a1 prints the full line")


def measure_with_uploads(args, eng, eng_mod, prepared, batch, params, local_rank, dist, torch):
    """Steady state with the H2D of every batch's rows inside the clock, overlapped (double buffered) and serial."""
    import threading
    from rpvg_amd import hip
    # (the rows, and the path side the device reads: PathInfo::group_id and PathInfo::source_ids — the haplotype columns of every
    # cluster are formed on the device behind the copy, every batch)
    arrays = [batch.cluster_row_off, batch.cluster_path_off, batch.row_count, batch.row_noise, batch.row_grp_off, batch.grp_prob,
              batch.grp_idx_off, batch.path_idx, batch.path_group_id, batch.path_source_off, batch.source_id]
    for a in arrays:
        if a.nbytes >= PAGE_LOCK_MIN_BYTES:
            hip.host_register(a)
    uploader = eng_mod.Engine(local_rank, uploader=True)
    slots = [prepared, eng.prepare(batch)]
    try:
        upload_s = []
        errors = []

        def upload(slot):
            try:
                upload_s.append(slot.reupload(uploader))
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        def step(k):
            t = threading.Thread(target=upload, args=(slots[(k + 1) % 2],))
            t.start()
            eng.run_raw(args.model, params, slots[k % 2])
            t.join()
            if errors:
                raise errors[0]

        slots[0].reupload(uploader)
        for k in range(max(2, args.warmup)):
            step(k)
        upload_s.clear()
        eng.reset_stats()
        barrier_sync(dist, torch)
        import resource

        threads0 = thread_cpu() if os.environ.get("RPVG_BENCH_THREAD_CPU") else None
        cpu0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        step_ms = []
        for k in range(args.steps):
            ts = time.perf_counter()
            step(k)
            step_ms.append((time.perf_counter() - ts) * 1e3)
        barrier_sync(dist, torch)
        overlapped = max_over_ranks(time.perf_counter() - t0, dist, torch)
        cpu1 = resource.getrusage(resource.RUSAGE_SELF)
        if threads0 is not None:
            print_thread_cpu(threads0, args.steps)
        host_cpu_ms = ((cpu1.ru_utime - cpu0.ru_utime) + (cpu1.ru_stime - cpu0.ru_stime)) * 1e3 / args.steps
        stats = eng.stats()
        t0 = time.perf_counter()
        for k in range(args.steps):  # no overlap: upload, then estimate
            slots[k % 2].reupload(uploader)
            eng.run_raw(args.model, params, slots[k % 2])
        serial = max_over_ranks(time.perf_counter() - t0, dist, torch)
    finally:
        slots[1].free()
        uploader.close()
        for a in arrays:
            if a.nbytes >= PAGE_LOCK_MIN_BYTES:
                hip.host_unregister(a)
    bytes_per_batch = float(sum(a.nbytes for a in arrays))
    up_ms = 1e3 * sum(upload_s) / max(1, len(upload_s))
    ordered = sorted(step_ms)
    spread = dict(median=ordered[len(ordered) // 2], min=ordered[0], max=ordered[-1], p10=ordered[len(ordered) // 10], p90=ordered[(9 * len(ordered)) // 10],
                  in_order=[round(x, 2) for x in step_ms] if len(step_ms) <= 64 else None,
                  note="wall time of the single steps of the timed region on this rank (ms_per_step is their mean between the barriers)")
    return dict(stats=stats, ms_per_step_with_h2d=overlapped / args.steps * 1e3, ms_per_step_with_h2d_serial=serial / args.steps * 1e3,
                ms_per_step_spread=spread, host_cpu_ms_per_step=host_cpu_ms,
                host_cpu_note="user + system CPU time of the process (every host thread: lanes, OpenMP teams, uploader) per step of the timed region, getrusage",
                h2d_ms_per_batch=up_ms, h2d_bytes_per_batch=bytes_per_batch, h2d_gb_per_s=bytes_per_batch / 1e9 / (up_ms / 1e3),
                h2d_note="rows of every batch uploaded from page-locked host arrays inside the clock: validation on the host, H2D, "
                         "expansion on the device; overlapped = two resident slots, uploader engine under the previous batch's kernels")


def dataset_parts(batch, fractions):
    """The batch's clusters cut into consecutive ranges holding the given fractions of its rows (ClusterRange: no copy of the rows)."""
    import numpy as np
    rows = batch.cluster_row_off.astype(np.float64) / max(1.0, float(batch.cluster_row_off[-1]))
    total = float(sum(fractions))
    cuts, run = [0], 0.0
    for f in fractions[:-1]:
        run += f / total
        cuts.append(max(cuts[-1] + 1, min(batch.num_clusters - (len(fractions) - len(cuts)), int(np.searchsorted(rows, run)))))
    cuts.append(batch.num_clusters)
    return [batch.cluster_range(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def measure_single_dataset(args, eng_mod, batch, params, local_rank, repeats=9):
    """ONE data set, once: what a real rpvg run hands over — 10 M read pairs in host memory — cut by cluster into parts that go
    through the batch pipeline one behind the other (the copy of part n + 1 under the kernels of part n, parts on engines of
    their own), wall clock from the first submit to the last estimate in its host container.  The pipeline is warm (contexts,
    memory pools), the data is not: every part is copied, validated and expanded inside the clock, and so is the cut itself (views
    of the caller's arrays and three small offset arrays per part).  Median and minimum of `repeats` passes."""
    from rpvg_amd import hip
    fractions = [float(x) for x in os.environ.get("RPVG_BENCH_PARTS", "0.5,0.5").split(",")]
    arrays = copied_arrays(batch)
    for a in arrays:
        if a.nbytes >= PAGE_LOCK_MIN_BYTES:
            hip.host_register(a)
    pipe = eng_mod.Pipeline(args.model, params, local_rank, workers=args.pipeline_workers)
    try:
        parts = dataset_parts(batch, fractions)
        for slot, part in enumerate(parts):
            pipe.prepare_slot(slot, part)
        times = []
        for rep in range(repeats + 2):
            t0 = time.perf_counter()
            cut = dataset_parts(batch, fractions)
            for slot, part in enumerate(cut):
                pipe.submit(part, slot, compact=True, narrow=NARROW_UPLOADS)
            pipe.wait()
            if rep >= 2:
                times.append((time.perf_counter() - t0) * 1e3)
        est = [e for slot in range(len(parts)) for e in pipe.result(slot)]
    finally:
        pipe.close()
        for a in arrays:
            if a.nbytes >= PAGE_LOCK_MIN_BYTES:
                hip.host_unregister(a)
    times.sort()
    return dict(ms=times[len(times) // 2], ms_min=times[0], ms_max=times[-1], repeats=repeats, parts=len(parts),
                rows_per_part=[int(p.num_rows) for p in parts], clusters_per_part=[int(p.num_clusters) for p in parts], estimates=est,
                note="one cold configs[2] data set (10 M read pairs in page-locked host memory) cut by cluster into parts through the batch pipeline: "
                     "first submit -> last estimate in its host container, the cut inside; median of the repeats")


def host_bound_line(args, eng_mod, batch, params, local_rank, ranks=8, steps=40):
    """The pipeline under the host budget ONE rank of an 8-rank node has: every thread the pipeline starts confined to the rank's
    share of the CPUs the cgroup grants (cpu.max; all hardware threads if it grants everything).  Against the unconfined
    step it is the weak-scaling efficiency the host side alone allows such a node (predicted; not a multi-GPU measurement)."""
    from rpvg_amd import hip
    quota = cpu_quota() or float(os.cpu_count() or 1)
    share = max(1, int(quota // ranks))
    allowed = sorted(os.sched_getaffinity(0))
    mine = set(allowed[:share])
    arrays = copied_arrays(batch)
    for a in arrays:
        if a.nbytes >= PAGE_LOCK_MIN_BYTES:
            hip.host_register(a)
    os.sched_setaffinity(0, mine)  # (threads started from here on inherit it: the pipeline's uploader and estimator threads)
    try:
        pipe = eng_mod.Pipeline(args.model, params, local_rank, workers=args.pipeline_workers)
        try:
            slots = pipe.workers + 4
            pipe.prepare_slots(batch, slots)
            for k in range(2 * slots):
                pipe.submit(batch, k % slots, compact=True, narrow=NARROW_UPLOADS)
            pipe.wait()
            t0 = time.perf_counter()
            for k in range(steps):
                pipe.submit(batch, k % slots, compact=True, narrow=NARROW_UPLOADS)
            pipe.wait()
            ms = (time.perf_counter() - t0) / steps * 1e3
        finally:
            pipe.close()
    finally:
        os.sched_setaffinity(0, set(allowed))
        for a in arrays:
            if a.nbytes >= PAGE_LOCK_MIN_BYTES:
                hip.host_unregister(a)
    return dict(ranks=ranks, node_cpus=quota, cpus_per_rank=share, ms_per_step=ms, steps=steps,
                note="the headline's pipeline with all of its threads confined (sched_setaffinity) to one rank's share of the node's CPUs: what the "
                     "host side alone allows a rank of an 8-rank node; efficiency_bound = unconfined ms_per_step / this")


def thread_cpu():
    """CPU seconds (user, system) of every thread of the process so far (RPVG_BENCH_THREAD_CPU=1: who burnt the host's time)."""
    import glob
    ticks = os.sysconf("SC_CLK_TCK")
    out = {}
    for stat in glob.glob("/proc/self/task/*/stat"):
        try:
            text = open(stat).read()
            fields = text[text.rindex(")") + 2:].split()
            out[stat.split("/")[4]] = (int(fields[11]) / ticks, int(fields[12]) / ticks)
        except Exception:  # noqa: BLE001
            pass
    return out


def print_thread_cpu(threads0, steps):
    threads1 = thread_cpu()
    rows = sorted(((sum(threads1[t]) - sum(threads0.get(t, (0.0, 0.0))), threads1[t][0] - threads0.get(t, (0.0, 0.0))[0],
                    threads1[t][1] - threads0.get(t, (0.0, 0.0))[1], t) for t in threads1), reverse=True)
    print("thread cpu over the timed region, ms per step (total user sys tid):", file=sys.stderr)
    for r in rows[:16]:
        try:
            name = open(f"/proc/self/task/{r[3]}/comm").read().strip()
        except OSError:
            name = "?"
        print(f"  {r[0] * 1e3 / steps:7.2f} {r[1] * 1e3 / steps:7.2f} {r[2] * 1e3 / steps:7.2f} {r[3]} {name}", file=sys.stderr)
    busy = [r for r in rows if r[0] > 0]
    print(f"  threads {len(rows)}, with CPU time {len(busy)}, sum {sum(r[0] for r in rows) * 1e3 / steps:.1f} ms per step "
          f"(user {sum(r[1] for r in rows) * 1e3 / steps:.1f}, system {sum(r[2] for r in rows) * 1e3 / steps:.1f}); "
          f"the 16 busiest {sum(r[0] for r in rows[:16]) * 1e3 / steps:.1f}", file=sys.stderr)


def copied_arrays(batch):
    """The host arrays of a batch that its copy to the GPU reads (to be page-locked): the two long offset arrays as counts of one
    byte where they fit, else in 32 bits, and the 16-bit path indices and source ids and one-byte read counts where those fit
    (include/rpvg_batch.h: what a caller that flattens rows for the GPU writes)."""
    offsets = batch.counts8() or batch.offsets32()
    forms = batch.narrow() if NARROW_UPLOADS else {}
    counts = [forms["row_count8"], forms["row_count_escape_row"], forms["row_count_escape_count"]] if "row_count8" in forms else [batch.row_count]
    noise = [forms["row_noise16"], forms["row_noise_table"]] if "row_noise16" in forms else [batch.row_noise]
    return noise + [batch.cluster_row_off, batch.cluster_path_off, offsets[0], batch.grp_prob,
            offsets[1], forms.get("path_idx16", batch.path_idx), batch.path_group_id, batch.path_source_off, forms.get("source_id16", batch.source_id)] + counts


PAGE_LOCK_MIN_BYTES = 1 << 16  # (smaller arrays — the list of the rows whose count does not fit a byte — go through the library's staging block)
NARROW_UPLOADS = not os.environ.get("RPVG_BENCH_WIDE_UPLOADS")  # (A/B: the 32-bit path indices, source ids and read counts of round 5)


def measure_pipeline(args, eng_mod, batch, params, local_rank, dist, torch):
    """K batches through BatchPipeline: submit() K times, wait().  The host arrays are page-locked; every batch is validated,
    copied and expanded on the device inside the clock (and its haplotype columns formed), several batches are in flight, the
    estimates of every batch land in host containers (PathClusterEstimates).  The pipeline is empty when the clock starts and
    when it stops: ramp-up and drain are inside."""
    import resource
    from rpvg_amd import hip
    arrays = copied_arrays(batch)
    for a in arrays:
        if a.nbytes >= PAGE_LOCK_MIN_BYTES:
            hip.host_register(a)
    pipe = eng_mod.Pipeline(args.model, params, local_rank, workers=args.pipeline_workers)
    try:
        slots = pipe.workers + 4
        pipe.prepare_slots(batch, slots)
        for k in range(max(args.warmup, 2 * slots)):
            pipe.submit(batch, k % slots, compact=True, narrow=NARROW_UPLOADS)
        pipe.wait()
        pipe.reset_stats()
        barrier_sync(dist, torch)
        threads0 = thread_cpu() if os.environ.get("RPVG_BENCH_THREAD_CPU") else None
        cpu0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        for k in range(args.steps):
            pipe.submit(batch, k % slots, compact=True, narrow=NARROW_UPLOADS)
        pipe.wait()
        barrier_sync(dist, torch)
        elapsed = max_over_ranks(time.perf_counter() - t0, dist, torch)
        cpu1 = resource.getrusage(resource.RUSAGE_SELF)
        if threads0 is not None:
            print_thread_cpu(threads0, args.steps)
        stats = pipe.stats()
        done = pipe.completions()
        workers = pipe.workers
        est = pipe.result((args.steps - 1) % slots)
    finally:
        pipe.close()
        for a in arrays:
            if a.nbytes >= PAGE_LOCK_MIN_BYTES:
                hip.host_unregister(a)
    host_cpu_ms = ((cpu1.ru_utime - cpu0.ru_utime) + (cpu1.ru_stime - cpu0.ru_stime)) * 1e3 / args.steps
    bytes_per_batch = float(sum(a.nbytes for a in arrays))
    # The pipeline's contexts time their EM launches only (rpvg_hip_ctx::span_level, context.hip: two events per span are two marker
    # commands between dependent kernels, and with every kernel family timed a batch takes 0.3-0.6 ms longer).  The per-family
    # device times, the busy union and the copies' device time of the line come from a second, shorter pass with all of them on.
    instrumented = stats
    if stats.get("busy_ms", 0.0) == 0.0 and not os.environ.get("RPVG_HIP_SPANS"):
        os.environ["RPVG_HIP_SPANS"] = "2"
        try:
            for a in arrays:
                if a.nbytes >= PAGE_LOCK_MIN_BYTES:
                    hip.host_register(a)
            pipe = eng_mod.Pipeline(args.model, params, local_rank, workers=args.pipeline_workers)
            try:
                steps2 = max(16, min(args.steps, 48))
                pipe.prepare_slots(batch, slots)
                for k in range(2 * slots):
                    pipe.submit(batch, k % slots, compact=True, narrow=NARROW_UPLOADS)
                pipe.wait()
                pipe.reset_stats()
                t2 = time.perf_counter()
                for k in range(steps2):
                    pipe.submit(batch, k % slots, compact=True, narrow=NARROW_UPLOADS)
                pipe.wait()
                instrumented_ms = (time.perf_counter() - t2) / steps2 * 1e3
                instrumented = pipe.stats()
                instrumented["steps"] = steps2
                instrumented["ms_per_step"] = instrumented_ms
            finally:
                pipe.close()
                for a in arrays:
                    if a.nbytes >= PAGE_LOCK_MIN_BYTES:
                        hip.host_unregister(a)
        finally:
            os.environ.pop("RPVG_HIP_SPANS", None)
        # (per step of ITS pass, scaled to the steps of the timed one: what the caller divides by)
        scale = args.steps / float(instrumented["steps"])
        for key in ("em_sparse_ms", "em_dense_ms", "loglik_ms", "build_ms", "h2d_ms", "collapse_ms", "busy_ms", "gibbs_ms", "search_tile_ms", "search_tile_launches"):
            if key in instrumented:
                stats[key] = instrumented[key] * scale
        for key in ("upload_copies_ms_per_batch", "upload_kernels_ms_per_batch"):
            if key in instrumented:
                stats[key] = instrumented[key]
    up_ms = stats.pop("upload_seconds_per_batch") * 1e3
    up_copies_ms, up_kernels_ms = stats.pop("upload_copies_ms_per_batch", None), stats.pop("upload_kernels_ms_per_batch", None)
    worker_ms = dict(finish_upload=stats.pop("worker_finish_ms_per_batch", None), estimate=stats.pop("worker_estimate_ms_per_batch", None),
                     wait_for_a_batch=stats.pop("worker_idle_ms_per_batch", None),
                     note="wall ms per batch of an estimator thread: the kernels behind the copies (rpvg_hip_batch_upload_finish), "
                          "estimateBatch, and waiting for the uploader")
    gaps = sorted((done[1:] - done[:-1]) * 1e3) if len(done) > 2 else []
    spread = None
    if gaps:
        spread = dict(median=gaps[len(gaps) // 2], min=gaps[0], max=gaps[-1], p10=gaps[len(gaps) // 10], p90=gaps[(9 * len(gaps)) // 10],
                      first_batch_done_ms=float(done[0]) * 1e3, last_batch_done_ms=float(done[-1]) * 1e3,
                      note="time between the completions of consecutive batches inside the timed region (ms_per_step = wall time between the "
                           "barriers / K, ramp-up of the first batch and drain of the last included)")
    instrumented_note = None
    if instrumented is not stats:
        instrumented_note = dict(steps=instrumented["steps"], ms_per_step=instrumented["ms_per_step"],
                                 note="a second pass of the pipeline with every kernel family timed (RPVG_HIP_SPANS=2): the source of the line's per-family "
                                      "device times (`kernels`, `roofline_search`), `gpu_active_frac` and the copies' device time; the timed pass times its EM "
                                      "launches only (`roofline`, `em_kernels`)")
        busy_frac = (instrumented.get("busy_ms", 0.0) / instrumented["steps"]) / instrumented["ms_per_step"] if instrumented["ms_per_step"] > 0 else None
        instrumented_note["gpu_active_frac"] = busy_frac
    return dict(stats=stats, instrumented_pass=instrumented_note, ms_per_step_with_h2d=elapsed / args.steps * 1e3, ms_per_step_spread=spread, host_cpu_ms_per_step=host_cpu_ms,
                host_cpu_note="user + system CPU time of the process (uploader, estimator threads, this thread) per step of the timed region, getrusage",
                pipeline=dict(workers=workers, resident_batches=workers + 1, estimates_slots=slots, worker_ms_per_batch=worker_ms,
                              note="rpvg_amd/host/batch_pipeline.hpp: one uploader thread (context of its own), `workers` estimator threads with a "
                                   "single-lane engine each; batches do not interact, results equal those of one call after the other "
                                   "(tests/test_hip_pipeline.py)"),
                h2d_ms_per_batch=up_ms, h2d_bytes_per_batch=bytes_per_batch, h2d_gb_per_s=bytes_per_batch / 1e9 / (up_ms / 1e3) if up_ms > 0 else None,
                h2d_copies_device_ms_per_batch=up_copies_ms, h2d_kernels_device_ms_per_batch=up_kernels_ms,
                h2d_note="rows and path side (PathInfo::group_id, source_ids) of every batch from page-locked host arrays inside the clock: offsets "
                         "checked on the host, H2D, validation + expansion + haplotype columns on the device",
                sample_estimates=est)


def run_a1(args, rank, local_rank, world, dist, torch):
    """The reference's own call pattern on the configs[2] workload: PathEstimator::estimate() once per cluster from an OpenMP team
    (src/main.cpp:829,976-977), clusters in the reference's order (src/main.cpp:811-827: descending read count, the order the
    generator emits).  A step = all clusters of the batch through estimate(); rows start on the host (ReadPathProbabilities
    objects, as the reference holds them) and every call flattens and uploads its own."""
    import numpy as np
    from rpvg_amd import synth, engine as eng_mod
    from rpvg_amd.batch import make_params
    params = make_params()
    K = max(8, int(round(5000 * args.scale)))
    total_paths = max(K, int(round(200000 * args.scale)))
    total_reads = int(round(10000000 * args.scale))
    batch = synth.generate(seed=3 + rank, num_clusters=K, total_paths=total_paths, total_reads=total_reads)
    eng = eng_mod.Engine(local_rank)
    t0 = time.perf_counter()
    prepared = eng.prepare(batch, per_cluster=True)  # the rows as vector<ReadPathProbabilities> per cluster, on the host
    prepare_s = time.perf_counter() - t0
    steps, warmup = max(1, min(args.steps, 10)), max(1, min(args.warmup, 2))
    for _ in range(warmup):
        eng.run_team(args.model, params, prepared, args.team, decode=False)
    barrier_sync(dist, torch)
    t0 = time.perf_counter()
    step_s = [eng.run_team(args.model, params, prepared, args.team, decode=False)[1] for _ in range(steps)]
    barrier_sync(dist, torch)
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, torch)
    est, _ = eng.run_team(args.model, params, prepared, args.team)
    whole, _ = eng.run(args.model, params, eng.prepare(batch))
    same = all(a.path_group_sets == b.path_group_sets and np.allclose(a.posteriors, b.posteriors, rtol=1e-9, atol=1e-12)
               and np.allclose(a.abundances, b.abundances, rtol=1e-9, atol=1e-9) and a.em_iters == b.em_iters for a, b in zip(est, whole))
    reads_all = sum_over_ranks(float(batch.total_reads), dist, torch)
    if rank != 0:
        return None
    ms = elapsed / steps * 1e3
    return dict(metric="read-pairs quantified/sec", value=reads_all / (ms / 1e3), unit="read-pairs/s", n_gpus=world, steps=steps, warmup=warmup,
                ms_per_step=ms, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                config=dict(workload=f"synthetic pantranscriptome: {total_reads} read pairs x {total_paths} paths in {K} clusters per GPU (BASELINE.json "
                                     f"configs[2]), -i {args.model}, reference defaults — through PathEstimator::estimate(), one call per cluster "
                                     f"from an OpenMP team of {args.team} threads (the reference's loop, src/main.cpp:829,976-977)",
                            team_threads=args.team, clusters_per_gpu=K),
                ms_per_step_in_order=[round(x * 1e3, 2) for x in step_s], estimates_equal_estimate_batch=bool(same), prepare_seconds=prepare_s,
                note="every call flattens its cluster on its own thread; the calls in flight are joined into batches of up to 256 clusters behind "
                     "the interface (PathEstimator::CallCombiner), up to three batches on the GPU at once; rows start on the host, uploads inside")


def gibbs_line(args, local_rank, steps=16):
    """BASELINE.json configs[4] (`-i haplotypes -y 2 --use-hap-gibbs`, 10M reads x 500k paths) inside the default run, so that the
    driver's record carries it: a short run of the --workload s5 bench as a child process (its own engines, its own memory)."""
    import subprocess
    env = dict(os.environ, RPVG_BENCH_NO_SINGLE="1")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "s5", "--steps", str(steps), "--warmup", "4", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-400:])
    d = json.loads(out.stdout.strip().splitlines()[-1])
    r = d["roofline"]
    keep = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "evals_per_step", "flops_per_eval", "ms_per_step",
                              "sampler_span_ms_per_step", "chains_and_bookkeeping_ms_per_step", "dominant_by_device_time", "note") if k in r}
    keep.update(workload=d["config"]["workload"], value=d["value"], unit_value="read-pairs/s", batch_ms_per_step=d["ms_per_step"], steps=d["steps"],
                gpu_active_frac=d.get("gpu_active_frac"), host_cpu_ms_per_step=d.get("host_cpu_ms_per_step"), mass_conserved=d.get("mass_conserved"),
                pipeline=d.get("pipeline", {}).get("workers"),
                line_note="the configs[4] workload through the same batch pipeline, every batch uploaded inside the clock; python bench.py --workload s5 "
                          "prints the full line")
    return keep


def dense_em_roofline(local_rank, rows=1000000, paths=2000, its=20):
    """Short measurement of the HBM-streaming dense EM kernel (BASELINE.json configs[1] shape) for the roofline
    record of the default run: the batched kernels of the default workload run out of L2/LDS."""
    from rpvg_amd import hip
    Cn = paths + 1
    ld = (Cn + 1) & ~1
    ctx = hip.Context(local_rank)
    d_P = d_c = None
    try:
        d_P, d_c = ctx.malloc(rows * ld * 8), ctx.malloc(rows * 8)
        ctx.synth_dense_cluster(2, rows, paths, d_P, ld, d_c)
        ctx.em_dense(d_P, rows, Cn, ld, d_c, float(rows), max_em_its=3, max_rel_em_conv=0.0)
        ctx.reset_stats()
        ctx.em_dense(d_P, rows, Cn, ld, d_c, float(rows), max_em_its=its, max_rel_em_conv=0.0)
        st = ctx.stats()
        ms = st["em_dense_ms"] / max(1, st["em_dense_launches"])
        nbytes = st["em_dense_alg_bytes"] / max(1, st["em_dense_launches"])
        achieved = (nbytes / 1e9) / (ms / 1e3)
        traffic = None
        pmc = load_pmc("pmc_traffic_c2.json")
        if pmc is not None:
            shape = pmc.get("shape", {})
            if shape.get("rows") == rows and shape.get("cols") == Cn:
                traffic = pmc.get("traffic_bytes_per_launch", pmc.get("traffic_bytes_per_step"))  # (tools/pmc_traffic.py calls a launch a step)
        return dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                    kernel="emDenseAccumWideKernel", ms_per_launch=ms, algorithmic_bytes_per_launch=nbytes,
                    workload=f"single dense cluster {rows} x {Cn} FP64 ({rows * Cn * 8 / 1e9:.1f} GB), {its} EM iterations; "
                             "one launch = one iteration's streaming pass")
    finally:
        if d_P:
            ctx.free(d_P)
        if d_c:
            ctx.free(d_c)
        ctx.close()


def run_rows(args, rank, local_rank, world, dist, torch):
    """Row construction (ReadPathProbabilities::addPathProbs + sort/merge, include/rpvg_rows.h) for the reads of the
    configs[2] workload: every rank builds the rows of its own batch (clusters are independent: no collective)."""
    import numpy as np
    from rpvg_amd import hip, synth
    from rpvg_amd.rows import RowParams
    K = max(8, int(round(5000 * args.scale)))
    total_paths = max(K, int(round(200000 * args.scale)))
    total_reads = int(round(10000000 * args.scale))
    batch, aligns = synth.generate_with_alignments(seed=3 + rank, num_clusters=K, total_paths=total_paths, total_reads=total_reads)
    # FragmentLengthDist(300, 50): log normal density per fragment length (src/fragment_length_dist.cpp:407-427)
    v = np.arange(65536, dtype=np.float64)
    frag = np.log(0.3989422804014327) - np.log(50.0) - 0.5 * ((v - 300.0) / 50.0) ** 2
    prm = RowParams(prob_precision=1e-8, min_noise_prob=0.0, is_single_end=False, frag_length_log_prob=frag)
    ctx = hip.Context(local_rank)
    t_up = time.perf_counter()
    dev = ctx.upload_alignments(aligns)  # inputs resident in HBM before the timed region (validation + H2D)
    upload_ms = (time.perf_counter() - t_up) * 1e3

    def step():
        rows = dev.build_rows(prm, merge=True)         # resident alignment lists -> resident merged rows
        hb = rows.to_batch_handle()                    # -> the batch the estimators take, still on the GPU
        return rows, hb

    def release(rows, hb):
        hip.lib().rpvg_hip_batch_free(ctx.handle, hb)
        rows.free()

    for _ in range(args.warmup):
        release(*step())
    barrier_sync(dist, torch)
    t0 = time.perf_counter()
    build_ms = merge_ms = 0.0
    for i in range(args.steps):
        drows, hb = step()
        if i + 1 < args.steps:
            release(drows, hb)
    barrier_sync(dist, torch)
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, torch)
    t_down = time.perf_counter()
    rows, build_ms, merge_ms = drows.download()
    download_ms = (time.perf_counter() - t_down) * 1e3
    release(drows, hb)
    build_ms *= args.steps
    merge_ms *= args.steps
    reads_all = sum_over_ranks(float(aligns.total_reads), dist, torch)
    if rank != 0:
        return None
    E = int(aligns.align_path_off[-1])
    A = int(aligns.read_align_off[-1])
    N = aligns.num_reads
    # algorithmic bytes of the row kernel: every input array once (4 B path index per entry, 16 B per alignment,
    # 17 B per read) + the padded row slices written once (12 B per group, 4 B per member, 16 B per row)
    alg_bytes = 4.0 * E + 16.0 * A + 17.0 * N + 12.0 * len(rows.grp_prob) + 4.0 * len(rows.path_idx) + 16.0 * N
    ach = (alg_bytes / 1e9) / (build_ms / args.steps / 1e3)
    same_structure = bool(np.array_equal(rows.cluster_row_off, batch.cluster_row_off) and np.array_equal(rows.row_count.sum(), batch.row_count.sum()))
    line = dict(
        metric="read-pairs through row construction/sec", value=reads_all / (elapsed / args.steps), unit="read-pairs/s", n_gpus=world,
        steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True, scaling="weak",
        vs_baseline=None, dtype="f64", data="synthetic",
        config=dict(workload=f"row construction for {total_reads} read pairs ({N} distinct alignment-path lists, {A} alignments, {E} path "
                             f"entries) x {total_paths} paths in {K} clusters per GPU (the reads of BASELINE.json configs[2]); alignment "
                             "lists resident in HBM -> merged rows resident in HBM as the estimators' batch", parallelism=f"clusters sharded, {world} rank(s), no collective"),
        roofline=dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None,
                      kernel="readRowKernel", ms_per_launch=build_ms / args.steps,
                      note="algorithmic bytes = inputs once + row slices once; the kernel is one wave per read and latency bound "
                           "(binary searches, sequential precision buckets), not a streaming kernel"),
        kernels=dict(row_kernels_ms_per_step=build_ms / args.steps, sort_merge_pack_ms_per_step=merge_ms / args.steps),
        rows_out=int(rows.num_rows), rows_match_generator=same_structure, upload_ms=upload_ms, download_ms=download_ms,
        value_including_upload_and_download=float(aligns.total_reads) / ((elapsed / args.steps) + (upload_ms + download_ms) / 1e3) * world)
    # path clustering (PathClusters) of the same reads: the paths of all alignments of a read are one
    # id set over global path ids (src/path_clusters.cpp:31-47); the components must refine the generator's clusters
    per_align = np.diff(aligns.align_path_off.astype(np.int64))
    align_read = np.repeat(np.arange(N), np.diff(aligns.read_align_off.astype(np.int64)))
    read_cluster = np.repeat(np.arange(K), np.diff(aligns.cluster_read_off.astype(np.int64)))
    entry_cluster = np.repeat(read_cluster[align_read], per_align)
    global_path = (aligns.align_path_idx.astype(np.int64) + aligns.cluster_path_off.astype(np.int64)[entry_cluster]).astype(np.uint32)
    ctx.reset_stats()
    t_cl = time.perf_counter()
    read_set_off = aligns.align_path_off[aligns.read_align_off.astype(np.int64)]
    p2c, members = ctx.path_clusters_flat(total_paths, read_set_off, global_path)
    cluster_ms = (time.perf_counter() - t_cl) * 1e3
    cl_stats = ctx.stats()
    path_home = np.repeat(np.arange(K), np.diff(aligns.cluster_path_off.astype(np.int64)))
    first_home = np.full(len(members), -1, dtype=np.int64)
    first_home[p2c[::-1]] = path_home[::-1]
    line["path_clustering"] = dict(ms=cluster_ms, device_span_ms=cl_stats["build_ms"] + cl_stats["h2d_ms"], paths=total_paths, id_sets=int(N), set_members=int(E), clusters_found=len(members),
                                   refines_generator_clusters=bool(np.array_equal(first_home[p2c], path_home)),
                                   note="host arrays in / out: rpvg_hip_path_clusters (union-find on the GPU); device_span_ms includes the pageable "
                                        "H2D copies, the union-find kernel itself is ~1.5 ms (profiles/r01/rocprofv3_rows_kernel_stats.csv)")
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        cores = pyoracle.max_threads()
        _, _, cl_secs = pyoracle.path_clusters_flat(total_paths, read_set_off, global_path)
        line["path_clustering"]["cpu_ms"] = cl_secs * 1e3
        _, secs = pyoracle.build_rows(aligns, prm, merge=True, num_threads=cores)
        line["cpu_baseline"] = dict(value=aligns.total_reads / secs, unit="read-pairs/s", cores=cores, kind="port",
                                    sample=f"the whole batch ({aligns.total_reads} read pairs, {secs:.2f} s): oracle/ addPathProbs + sort/merge, "
                                           f"OpenMP dynamic over clusters, {cores} threads" + cpu_quota_note())
    return line


def run_e2e(args, rank, local_rank, world, dist, torch):
    """The widened path in one pass: alignment-path lists resident in HBM -> merged rows (rpvg_hip_read_rows_build) -> the
    estimators' batch (rpvg_hip_read_rows_to_batch) -> -i haplotype-transcripts, for the reads of BASELINE.json configs[2]."""
    import numpy as np
    from rpvg_amd import engine as eng_mod, synth
    from rpvg_amd.batch import make_params
    from rpvg_amd.rows import RowParams
    params = make_params()
    K = max(8, int(round(5000 * args.scale)))
    total_paths = max(K, int(round(200000 * args.scale)))
    total_reads = int(round(10000000 * args.scale))
    batch, aligns = synth.generate_with_alignments(seed=3 + rank, num_clusters=K, total_paths=total_paths, total_reads=total_reads)
    eng = eng_mod.Engine(local_rank)
    frag = (300.0, 50.0, 0.0, 10)
    prepared = eng.prepare_from_alignments(aligns, batch, frag=frag, min_noise_prob=0.0)  # lists resident before the timed region
    for _ in range(args.warmup):
        eng.run_from_alignments_raw(args.model, params, prepared)
    barrier_sync(dist, torch)
    t0 = time.perf_counter()
    rows_s = est_s = 0.0
    for _ in range(args.steps):
        r, e = eng.run_from_alignments_raw(args.model, params, prepared)
        rows_s += r
        est_s += e
    barrier_sync(dist, torch)
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, torch)
    reads_all = sum_over_ranks(float(aligns.total_reads), dist, torch)
    est, _ = eng.run(args.model, params, prepared)
    mass_ok = all(abs(e.abundances.sum() + e.noise_count - e.total_count) <= 1e-6 * max(1.0, e.total_count) for e in est)
    if rank != 0:
        return None
    line = dict(
        metric="read-pairs quantified/sec from alignment-path lists", value=reads_all / (elapsed / args.steps), unit="read-pairs/s",
        n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True,
        scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
        config=dict(workload=f"alignment-path lists of {total_reads} read pairs ({aligns.num_reads} distinct lists) x {total_paths} paths in "
                             f"{K} clusters per GPU (the reads of BASELINE.json configs[2]) -> rows -> -i {args.model}, all on the GPU",
                    parallelism=f"clusters sharded, {world} rank(s), no collective"),
        stages=dict(row_construction_ms_per_step=rows_s / args.steps * 1e3, estimates_ms_per_step=est_s / args.steps * 1e3),
        mass_conserved=bool(mass_ok))
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        cores = pyoracle.max_threads()
        v = np.arange(65536, dtype=np.float64)
        prm = RowParams(prob_precision=1e-8, min_noise_prob=0.0, is_single_end=False,
                        frag_length_log_prob=pyoracle.frag_length_table(*frag))
        rows_o, secs_rows = pyoracle.build_rows(aligns, prm, merge=True, num_threads=cores)
        for name in ("path_group_id", "path_source_count", "path_source_off", "source_id", "path_effective_length"):
            setattr(rows_o, name, getattr(batch, name).copy())
        _, secs_est = pyoracle.run(args.model, params, rows_o, cores)
        line["cpu_baseline"] = dict(value=aligns.total_reads / (secs_rows + secs_est), unit="read-pairs/s", cores=cores, kind="port",
                                    sample=f"the whole batch: oracle row construction {secs_rows:.2f} s + estimates {secs_est:.2f} s, "
                                           f"OpenMP dynamic over clusters, {cores} threads" + cpu_quota_note())
    return line


def run_c2(args, rank, local_rank, world, dist, torch):
    """1M x 2k dense single cluster (BASELINE.json configs[1]).  Weak scaling (default): one replica of the
    cluster per GPU, no collective.  --scaling strong: the rows of ONE cluster are spread over the ranks and every
    EM iteration all-reduces its C partial column sums over RCCL (rpvg_hip_em_dense_sharded)."""
    import numpy as np
    from rpvg_amd import hip, dist as rdist
    R_all = max(1024, int(round(1000000 * args.scale)))
    N = 2000
    Cn = N + 1
    ld = (Cn + 1) & ~1
    its = 50
    sharded = args.scaling == "strong" and dist is not None
    ctx = hip.Context(local_rank)
    eng = prep = None
    d_P = d_c = None
    if sharded:
        # the rows of ONE cluster over the ranks: the raw ABI (rpvg_hip_em_dense_sharded), a matrix synthesised per rank
        rdist.init_engine_comm(ctx, dist, f"cuda:{local_rank}")
        r0, r1 = rdist.row_shard(R_all, rank, world)
        R, total = r1 - r0, float(R_all)
        d_P, d_c = ctx.malloc(R * ld * 8), ctx.malloc(R * 8)
        ctx.synth_dense_rows(2, r0, R, N, d_P, ld, d_c)

        def step():
            return ctx.em_dense(d_P, R, Cn, ld, d_c, total, max_em_its=its, max_rel_em_conv=0.0, sharded=True)
        stats_of = ctx
    else:
        # Through the estimator class: the cluster is a resident batch (its rows generated on the device), and
        # PathAbundanceEstimator::estimateBatch -> rpvg_hip_em_solve sends its one EM problem to the dense route of the
        # whole-GPU EM (rpvg_amd/csrc/em_grid.hip): the rows normalised straight into the dense matrix (fillDenseRowsKernel; RPVG_HIP_NO_FUSED_DENSE=1: compacted CSR, then a dense copy), `its` iterations.
        from rpvg_amd import engine as eng_mod
        from rpvg_amd.batch import make_params
        r0, R, total = 0, R_all, float(R_all)
        eng = eng_mod.Engine(local_rank)
        prep = eng.prepare_synth_dense(2 + rank, R, N)
        params = make_params(max_em_its=its, max_rel_em_conv=0.0)

        def step():
            eng.run_raw("transcripts", params, prep)
        stats_of = eng
    for _ in range(args.warmup):
        step()
    stats_of.reset_stats()
    barrier_sync(dist, torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier_sync(dist, torch)
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, torch)
    stats = stats_of.stats()
    if sharded:
        ab, noise, done = out
    else:
        est, _ = eng.run("transcripts", params, prep)
        ab, noise, done = est[0].abundances, est[0].noise_count, est[0].em_iters[0]
    reads_all = sum_over_ranks(float(R), dist, torch)
    if rank != 0:
        return None
    it_ms = stats["em_dense_ms"] / max(1, stats["em_dense_launches"])  # HIP events around the streaming-pass launches
    it_bytes = stats["em_dense_alg_bytes"] / max(1, stats["em_dense_launches"])
    achieved = (it_bytes / 1e9) / (it_ms / 1e3)
    traffic = None
    pmc = load_pmc("pmc_traffic_c2.json")
    if pmc is not None:
        shape = pmc.get("shape", {})
        if shape.get("rows") == R and shape.get("cols") == Cn:  # per launch of THIS rank's shard
            # separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950 note, + WRITE_SIZE); tools/pmc_traffic.py
            # calls a launch a step
            traffic = pmc.get("traffic_bytes_per_launch", pmc.get("traffic_bytes_per_step"))
    line = dict(
        metric="read-pairs quantified/sec", value=reads_all / (elapsed / args.steps), unit="read-pairs/s", n_gpus=world,
        steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True,
        scaling="strong" if sharded else "weak", vs_baseline=None, dtype="f64", data="synthetic",
        config=dict(workload=f"single dense cluster {R_all} read pairs x {N} paths (BASELINE.json configs[1]), -i transcripts EM, "
                             f"fixed budget of {its} EM iterations per step"
                             + ("" if sharded else ", through PathAbundanceEstimator::estimateBatch on a resident cluster batch"),
                    parallelism=(f"rows of one cluster spread over {world} rank(s), all-reduce of {Cn} doubles per EM iteration over RCCL"
                                 if sharded else f"replicas only, {world} rank(s)")),
        roofline=dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                      kernel="emDenseAccumWideKernel", ms_per_launch=it_ms, algorithmic_bytes_per_launch=it_bytes,
                      note="one launch = one EM iteration's streaming pass: 8*R*C (matrix, read once) + 8*R (counts) + 16*C bytes; "
                           "traffic = HBM bytes per launch from rocprofv3 PMC passes (profiles/%s/pmc_traffic_c2.json)" % PROFILE_ROUND + ""),
        em_iterations_per_step=int(done), mass_conserved=bool(abs(ab.sum() + noise - total) <= 1e-6 * total),
        step_breakdown_ms=dict(streaming_passes=stats["em_dense_ms"] / args.steps, build_and_compaction=stats["build_ms"] / args.steps,
                               note="streaming_passes = HIP-event spans of the iterations' streaming-pass launches; build_and_compaction = the "
                                    "spans of the kernels in front of them (the rows counted and normalised straight into the dense matrix: fillDenseRowsKernel); the rest of a step is "
                                    "the reduce / update / control launches and the host's waits between chunks of iterations"))
    if not args.no_cpu_baseline:
        # reference-shaped CPU EM on a row sample of the same matrix, one core (a single cluster is serial
        # in the reference: SURVEY.md F4)
        from oracle import pyoracle
        rows = min(R, 20000)
        if d_P is None:  # (the estimator route holds the cluster as a batch: the same rows as a dense matrix for the sample)
            s_P, s_c = ctx.malloc(rows * ld * 8), ctx.malloc(rows * 8)
            ctx.synth_dense_rows(2 + rank, 0, rows, N, s_P, ld, s_c)
            P = ctx.d2h(s_P, (rows, ld))[:, :Cn].copy()
            ctx.free(s_P)
            ctx.free(s_c)
        else:
            P = ctx.d2h(d_P, (rows, ld))[:, :Cn].copy()
        _, _, _, its_done, secs = pyoracle.em_dense(P, np.ones(rows), max_em_its=10, max_rel_em_conv=0.0)
        per_row_iter = secs / (rows * its_done)
        line["cpu_baseline"] = dict(value=1.0 / (per_row_iter * its), unit="read-pairs/s", cores=1, kind="port",
                                    sample=f"first {rows} rows of the same matrix, {its_done} EM iterations, {secs:.2f} s on one core; "
                                           f"scaled to {its} iterations per read pair")
    if d_P is not None:
        ctx.free(d_P)
        ctx.free(d_c)
    if prep is not None:
        prep.free()
        eng.close()
    ctx.close()
    return line


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    rank, local_rank, world, dist, torch = dist_setup(args.gpus)
    if DEVICE == "cuda" and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    runner = dict(c2=run_c2, rows=run_rows, e2e=run_e2e, a1=run_a1).get(args.workload, run_s3)
    line = runner(args, rank, local_rank, world, dist, torch)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes a version banner to the C library's stdout, which a pipe only sees when the buffer is flushed — at exit,
    # behind the JSON line, if nothing is done: flushed here, so that the JSON line is the last thing on stdout
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    if rank == 0:
        assert line["n_gpus"] == args.gpus, (line["n_gpus"], args.gpus)
        if NUMA_BINDING is not None:
            line["numa_binding_rank0"] = NUMA_BINDING
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
