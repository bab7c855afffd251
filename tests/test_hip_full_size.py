"""GPU runs at BASELINE.json's own sizes.

configs[0] (100k read pairs x 36,120 paths in 2,177 clusters, `-i transcripts`) and configs[2] (10M read pairs x
200k paths in 5,000 clusters, `-i haplotype-transcripts`) are small enough for the CPU oracle on a many-core
host, so they are compared with it cluster by cluster; on top of that the size-independent properties of the
domain are asserted: the writers' mass invariant (src/threaded_output_writer.cpp:327-328), posteriors of a
transcript's group sets summing to at most one, idempotence of a second run.  configs[1] (one dense 1M x 2001
cluster, 16 GB) is beyond the oracle; it is pinned by a closed form — from the uniform start the first EM
iteration yields the column means — plus determinism, mass conservation and sharded == unsharded.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle
from rpvg_amd import engine as eng_mod, hip, synth
from rpvg_amd.batch import make_params
from tests import small_cases

pytestmark = pytest.mark.gpu

REL = 1e-6


@pytest.fixture(scope="module")
def engine():
    e = eng_mod.Engine(0)
    yield e
    e.close()


def _oracle_threads():
    return max(1, min(pyoracle.max_threads(), os.cpu_count() or 1))


def _compare_all(got, ref):
    assert len(got) == len(ref)
    for k, (g, r) in enumerate(zip(got, ref)):
        gk, rk = g.keyed(), r.keyed()
        assert set(gk) == set(rk), f"cluster {k}: group sets differ"
        for key, (post, ab) in rk.items():
            assert small_cases.rel_close(gk[key][0], post, rel=REL, floor=1e-8), (k, key)
            assert small_cases.rel_close(gk[key][1], ab, rel=REL), (k, key)
        assert g.total_count == r.total_count, k
        assert abs(g.noise_count - r.noise_count) <= REL * max(1.0, r.total_count), k
        assert dict(zip(g.em_cols, g.em_iters)) == dict(zip(r.em_cols, r.em_iters)), f"cluster {k}: EM iterations differ"


def _mass_conserved(estimates):
    for k, e in enumerate(estimates):
        if e.total_count > 0:
            assert abs(e.abundances.sum() + e.noise_count - e.total_count) <= 1e-9 * e.total_count, k


def test_config0_transcripts_100k_reads_matches_oracle(engine):
    batch = synth.generate(seed=1, num_clusters=2177, total_paths=36120, total_reads=100000)
    assert batch.num_clusters == 2177 and int(batch.row_count.sum()) == 100000
    params = make_params()
    ref, _ = pyoracle.run("transcripts", params, batch, _oracle_threads())
    got, _ = engine.run("transcripts", params, engine.prepare(batch))
    _compare_all(got, ref)
    _mass_conserved(got)
    assert sum(e.total_count for e in got) == 100000


@pytest.fixture(scope="module")
def config2_batch():
    return synth.generate(**synth.FULL)


def test_config2_haplotype_transcripts_10m_reads_matches_oracle(engine, config2_batch):
    batch = config2_batch
    assert batch.num_clusters == 5000 and int(batch.row_count.sum()) == 10000000
    params = make_params()
    prep = engine.prepare(batch)
    got, _ = engine.run("haplotype-transcripts", params, prep)
    ref, _ = pyoracle.run("haplotype-transcripts", params, batch, _oracle_threads())
    _compare_all(got, ref)
    _mass_conserved(got)
    assert sum(e.total_count for e in got) == 10000000

    # posteriors of the group sets of one transcript (same group_id) sum to at most one
    first_path_group = [batch.path_group_id[int(batch.cluster_path_off[k]):int(batch.cluster_path_off[k + 1])]
                        for k in range(batch.num_clusters)]
    for k, e in enumerate(got):
        per_transcript = {}
        for s, p in zip(e.path_group_sets, e.posteriors):
            t = int(first_path_group[k][s[0]])
            per_transcript[t] = per_transcript.get(t, 0.0) + float(p)
        assert all(v <= 1 + 1e-9 for v in per_transcript.values()), k

    # idempotence: a second run on the same resident batch gives the same sets, posteriors, abundances and iteration counts bit
    # for bit (the log-likelihood kernels are deterministic, and the EM's column sums have one order of additions: an accumulator
    # vector per wavefront, added up in wavefront order — em_sparse.hip, emSparseProblem) — and so does a third
    for _ in range(2):
        again, _ = engine.run("haplotype-transcripts", params, prep)
        for k, (a, b) in enumerate(zip(got, again)):
            assert a.path_group_sets == b.path_group_sets, k
            assert np.array_equal(a.posteriors, b.posteriors), k
            assert np.array_equal(a.abundances, b.abundances), k
            assert a.noise_count == b.noise_count, k
            assert list(a.em_iters) == list(b.em_iters), k


def test_config2_transcripts_and_haplotypes_at_full_size_conserve_mass(engine, config2_batch):
    params = make_params()
    prep = engine.prepare(config2_batch)
    got, _ = engine.run("transcripts", params, prep)
    _mass_conserved(got)
    assert sum(e.total_count for e in got) == 10000000
    hap, _ = engine.run("haplotypes", params, prep)
    for k, e in enumerate(hap):
        if e.total_count > 0 and len(e.posteriors):
            assert abs(e.posteriors.sum() - 1) <= 1e-9, k  # Bounded normalises over the kept pairs


def test_config1_dense_cluster_16gb():
    R, N = 1000000, 2000
    Cn = N + 1
    ld = (Cn + 1) & ~1
    ctx = hip.Context(0)
    d_P = d_c = None
    try:
        d_P, d_c = ctx.malloc(R * ld * 8), ctx.malloc(R * 8)
        ctx.synth_dense_cluster(2, R, N, d_P, ld, d_c)

        # closed form: a0 uniform and every row summing to one => s_i = a0, a'_j = sum_i c_i P_ij / T = column mean
        col_sum = np.zeros(Cn)
        row_err = 0.0
        chunk = 50000
        for r0 in range(0, R, chunk):
            part = ctx.d2h(d_P + r0 * ld * 8, (chunk, ld))[:, :Cn]
            col_sum += part.sum(axis=0)
            row_err = max(row_err, float(np.abs(part.sum(axis=1) - 1).max()))
        assert row_err < 1e-12
        ab1, noise1, its1 = ctx.em_dense(d_P, R, Cn, ld, d_c, float(R), max_em_its=1, max_rel_em_conv=0.0)
        assert its1 == 1
        keep = col_sum[:N] / R >= 1e-8  # sub-threshold components are zeroed and moved to noise (:100-113)
        assert np.allclose(ab1[keep], col_sum[:N][keep], rtol=1e-9)
        assert np.all(ab1[~keep] == 0)
        assert abs(ab1.sum() + noise1 - R) <= 1e-9 * R

        ab, noise, its = ctx.em_dense(d_P, R, Cn, ld, d_c, float(R), max_em_its=12, max_rel_em_conv=0.0)
        ab2, noise2, its2 = ctx.em_dense(d_P, R, Cn, ld, d_c, float(R), max_em_its=12, max_rel_em_conv=0.0)
        assert its == its2 == 12
        assert np.array_equal(ab, ab2) and noise == noise2  # deterministic reduction order
        assert abs(ab.sum() + noise - R) <= 1e-9 * R

        ctx.comm_init(hip.Context.comm_unique_id(), 1, 0)
        ab3, noise3, its3 = ctx.em_dense(d_P, R, Cn, ld, d_c, float(R), max_em_its=12, max_rel_em_conv=0.0, sharded=True)
        assert its3 == 12 and np.array_equal(ab, ab3) and noise == noise3
    finally:
        for d in (d_P, d_c):
            if d:
                ctx.free(d)
        ctx.close()


def test_config1_dense_cluster_through_the_estimator_class(engine):
    """BASELINE.json configs[1] behind `-i transcripts`: the 1M x 2 000 cluster is a resident cluster batch (rows generated on
    the device), PathAbundanceEstimator::estimateBatch (src/path_abundance_estimator.cpp:18-45) solves it through
    rpvg_hip_em_solve, whose size bin sends the problem to the dense route of the whole-GPU EM.  Same property checks as the
    raw-ABI test above: the closed form of the first iteration (against the column sums of the same cluster synthesised as
    a dense matrix), determinism, the writers' mass invariant."""
    R, N = 1000000, 2000
    Cn = N + 1
    ld = (Cn + 1) & ~1
    prep = engine.prepare_synth_dense(2, R, N)
    ctx = hip.Context(0)
    d_P = d_c = None
    try:
        chunk = 50000
        d_P, d_c = ctx.malloc(chunk * ld * 8), ctx.malloc(chunk * 8)
        col_sum = np.zeros(Cn)
        for r0 in range(0, R, chunk):
            ctx.synth_dense_rows(2, r0, chunk, N, d_P, ld, d_c)
            col_sum += ctx.d2h(d_P, (chunk, ld))[:, :Cn].sum(axis=0)

        engine.reset_stats()
        got1, _ = engine.run("transcripts", make_params(max_em_its=1, max_rel_em_conv=0.0), prep)
        e1 = got1[0]
        assert e1.em_iters == [1] and e1.total_count == R
        keep = col_sum[:N] / R >= 1e-8  # sub-threshold components are zeroed and moved to noise (:100-113)
        assert np.allclose(e1.abundances[keep], col_sum[:N][keep], rtol=1e-9)
        assert np.all(e1.abundances[~keep] == 0)
        assert abs(e1.abundances.sum() + e1.noise_count - R) <= 1e-9 * R
        st = engine.stats()
        assert st["em_dense_launches"] == 1 and st["em_kernel"]["emGridAccumKernel"]["problems"] == 1

        params = make_params(max_em_its=12, max_rel_em_conv=0.0)
        a, _ = engine.run("transcripts", params, prep)
        b, _ = engine.run("transcripts", params, prep)
        assert a[0].em_iters == [12] and b[0].em_iters == [12]
        assert np.array_equal(a[0].abundances, b[0].abundances) and a[0].noise_count == b[0].noise_count  # fixed reduction orders
        assert abs(a[0].abundances.sum() + a[0].noise_count - R) <= 1e-9 * R

        # and the raw ABI on the dense matrix of the same cluster's first rows agrees with the batch route on those rows:
        # (a sub-cluster through both doors)
        sub = 20000
        ctx.synth_dense_rows(2, 0, sub, N, d_P, ld, d_c)
        ab_raw, noise_raw, its_raw = ctx.em_dense(d_P, sub, Cn, ld, d_c, float(sub), max_em_its=12, max_rel_em_conv=0.0)
        prep_sub = engine.prepare_synth_dense(2, sub, N)
        try:
            c, _ = engine.run("transcripts", params, prep_sub)
        finally:
            prep_sub.free()
        assert its_raw == 12 and c[0].em_iters == [12]
        assert np.allclose(c[0].abundances, ab_raw, rtol=1e-9, atol=1e-9) and abs(c[0].noise_count - noise_raw) <= 1e-9 * sub
    finally:
        for d in (d_P, d_c):
            if d:
                ctx.free(d)
        ctx.close()
        prep.free()


def test_config4_diploid_haplotype_gibbs_10m_reads_follows_the_reference_stream(engine):
    """BASELINE.json configs[4]: -i haplotypes -y 2 --use-hap-gibbs on 10M reads x 500k paths.  The sampler keeps
    the reference's mt19937(rng_seed + i) per cluster and libstdc++ distributions, so the sampled diplotypes and their
    frequencies must equal the oracle's exactly, cluster by cluster."""
    batch = synth.generate(seed=5, num_clusters=5000, total_paths=500000, total_reads=10000000, max_cluster_paths=5000)
    assert batch.num_paths == 500000 and int(batch.row_count.sum()) == 10000000
    params = make_params(use_hap_gibbs=1, rng_seed=11)
    got, _ = engine.run("haplotypes", params, engine.prepare(batch))
    ref, _ = pyoracle.run("haplotypes", params, batch, _oracle_threads())
    assert len(got) == len(ref)
    for k, (g, r) in enumerate(zip(got, ref)):
        assert g.path_group_sets == r.path_group_sets, k  # same sets in the same first-visit order
        assert np.allclose(g.posteriors, r.posteriors, rtol=0, atol=1e-12), k
        assert g.total_count == r.total_count, k
        if r.total_count > 0 and len(r.posteriors):
            assert abs(g.posteriors.sum() - 1) <= 1e-9, k


def test_config3_clusters_sharded_over_a_device_group_at_full_size(engine, config2_batch):
    """BASELINE.json configs[3]: the configs[2] workload with its 5 000 clusters sharded over the GPUs of a node
    (rpvg_amd/host/device_group.hpp: clusters bin-packed longest first, one engine and one host thread per entry, final
    abundance gather + TPM denominator).  A one-GPU box lists its GPU twice: two shards side by side — the bin packing, the
    per-engine thread budget and the gather see all 5 000 clusters.  Every cluster must get what it gets from one engine
    on the whole batch (sets, posteriors and EM iteration counts identical; abundances up to the summation order of the EM's
    LDS atomics), and the gather must return every cluster's abundances in cluster order."""
    from rpvg_amd import dist as rdist
    batch = config2_batch
    params = make_params()
    whole, _ = engine.run("haplotype-transcripts", params, engine.prepare(batch))
    group = eng_mod.DeviceGroup([0, 0])
    try:
        got, secs = group.run("haplotype-transcripts", params, batch)
        assert secs > 0 and len(got) == batch.num_clusters
        for k, (g, w) in enumerate(zip(got, whole)):
            assert g.path_group_sets == w.path_group_sets, k
            assert np.array_equal(g.posteriors, w.posteriors), k
            assert np.allclose(g.abundances, w.abundances, rtol=1e-9, atol=1e-9), k
            assert g.total_count == w.total_count and abs(g.noise_count - w.noise_count) <= 1e-9 * max(1.0, w.total_count), k
            assert g.em_iters == w.em_iters and g.em_cols == w.em_cols, k
        where = group.partition()
        costs = rdist.cluster_costs(batch)
        assert [sorted(np.nonzero(where == d)[0].tolist()) for d in range(2)] == rdist.partition_clusters(costs, 2)
        shares = [float(costs[where == d].sum()) for d in range(2)]
        assert max(shares) <= 1.02 * (sum(shares) / 2)  # longest-first packing of 5 000 clusters: within 2 % of even
        flat = np.concatenate([g.abundances for g in got])
        gathered, tpm_den = group.gather(len(flat))
        assert np.array_equal(gathered, flat)
        assert abs(tpm_den - rdist.local_transcript_count(got, batch)) <= 1e-9 * tpm_den
    finally:
        group.close()
