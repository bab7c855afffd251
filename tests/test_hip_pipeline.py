"""The widened path end to end on the GPU: alignment-path lists over GLOBAL path ids -> path clusters
(rpvg_hip_path_clusters) -> per-cluster alignment batch in the reference's cluster order (src/main.cpp:811-827)
-> merged rows (rpvg_hip_read_rows_build) -> estimates, against the same pipeline on the CPU oracle."""
import numpy as np
import pytest

from oracle import pyoracle
from rpvg_amd import engine as eng_mod, synth
from rpvg_amd.batch import ClusterBatch, make_params
from rpvg_amd.rows import AlignmentBatch, RowParams
from tests import small_cases

pytestmark = pytest.mark.gpu


def _global_reads(batch, aligns):
    """Every read of the generated batch as (count, mapq, noise score, [(score, align length, frag length, [global path ids])])."""
    reads = []
    cpo = aligns.cluster_path_off.astype(np.int64)
    for k in range(aligns.num_clusters):
        for r in range(int(aligns.cluster_read_off[k]), int(aligns.cluster_read_off[k + 1])):
            al = []
            for a in range(int(aligns.read_align_off[r]), int(aligns.read_align_off[r + 1])):
                idx = aligns.align_path_idx[int(aligns.align_path_off[a]):int(aligns.align_path_off[a + 1])].astype(np.int64) + cpo[k]
                al.append((int(aligns.align_score_sum[a]), int(aligns.align_length[a]), int(aligns.align_frag_length[a]), [int(x) for x in idx]))
            reads.append((int(aligns.read_count[r]), int(aligns.read_min_mapq[r]), int(aligns.read_noise_score[r]), al))
    return reads


def _cluster_batches(reads, path_info, p2c, members):
    """Reads and paths regrouped by path cluster, clusters ordered as the reference orders them (descending number of
    alignment lists, ties by descending cluster index: src/main.cpp:811-827); returns (AlignmentBatch, path ClusterBatch)."""
    per_cluster = [[] for _ in members]
    for rd in reads:
        per_cluster[int(p2c[rd[3][0][3][0]])].append(rd)
    order = sorted(range(len(members)), key=lambda c: (-len(per_cluster[c]), -c))
    clusters, path_clusters = [], []
    for c in order:
        local = {p: i for i, p in enumerate(members[c])}
        paths = [path_info[p] for p in members[c]]
        rds = [dict(count=cnt, min_mapq=mq, noise_score=ns,
                    aligns=[(s, al, fl, sorted(local[p] for p in ids)) for (s, al, fl, ids) in aligns])
               for (cnt, mq, ns, aligns) in per_cluster[c]]
        clusters.append(dict(paths=[dict(effective_length=p["effective_length"], source_count=p["source_count"]) for p in paths], reads=rds))
        path_clusters.append(dict(paths=paths, rows=[]))
    return AlignmentBatch.from_clusters(clusters), ClusterBatch.from_clusters(path_clusters)


def test_alignments_to_estimates_on_the_gpu(hip_ctx):
    batch, aligns = synth.generate_with_alignments(seed=17, num_clusters=30, total_paths=700, total_reads=40000)
    reads = _global_reads(batch, aligns)
    rng = np.random.default_rng(3)
    rng.shuffle(reads)  # the clustering must not depend on the order of the reads
    path_info = [p for k in range(batch.num_clusters) for p in batch.cluster(k)["paths"]]
    # one id set per read: the reference links the paths of ALL alignments of a list to the first path of its
    # first alignment (src/path_clusters.cpp:31-47)
    id_sets = [[p for (_, _, _, ids) in rd[3] for p in ids] for rd in reads]

    p2c, members = hip_ctx.path_clusters(len(path_info), id_sets)
    p2c_o, members_o, _ = pyoracle.path_clusters(len(path_info), id_sets)
    assert np.array_equal(p2c, p2c_o) and members == members_o
    home = np.repeat(np.arange(batch.num_clusters), np.diff(batch.cluster_path_off.astype(np.int64)))
    assert all(len({int(home[p]) for p in m}) == 1 for m in members)  # components refine the generator's clusters

    al_batch, path_batch = _cluster_batches(reads, path_info, p2c, members)
    frag = pyoracle.frag_length_table(300.0, 50.0, 0.0, 10)
    prm = RowParams(min_noise_prob=0.0, frag_length_log_prob=frag)

    # CPU pipeline: oracle rows -> oracle estimates
    rows_o, _ = pyoracle.build_rows(al_batch, prm, merge=True)
    for name in ("path_group_id", "path_source_count", "path_source_off", "source_id", "path_effective_length"):
        setattr(rows_o, name, getattr(path_batch, name).copy())
    want, _ = pyoracle.run("haplotype-transcripts", make_params(), rows_o, 2)

    # GPU pipeline: rows built on the device and handed to the estimators without leaving it
    e = eng_mod.Engine(0)
    try:
        prep = e.prepare_from_alignments(al_batch, path_batch, frag=(300.0, 50.0, 0.0, 10), min_noise_prob=0.0)
        got, _ = e.run("haplotype-transcripts", make_params(), prep)
    finally:
        e.close()

    assert len(got) == len(want) == len(members)
    assert sum(g.total_count for g in got) == 40000
    for g, w in zip(got, want):
        gk, wk = g.keyed(), w.keyed()
        assert set(gk) == set(wk)
        assert g.total_count == w.total_count
        for key, (post, ab) in wk.items():
            assert abs(gk[key][0] - post) <= 1e-6 * max(abs(post), 1e-8) + 1e-8
            assert np.allclose(gk[key][1], ab, rtol=1e-6, atol=1e-8)
        assert dict(zip(g.em_cols, g.em_iters)) == dict(zip(w.em_cols, w.em_iters))


# ---- the GPUs of a node behind one call (rpvg_amd/host/device_group.hpp) ---------------------------------------------

@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]], ids=["one", "two-shards", "three-shards"])
@pytest.mark.parametrize("model", ["transcripts", "haplotype-transcripts"])
def test_device_group_shards_clusters_and_gathers(model, devices):
    """One engine and one host thread per entry (a one-GPU box lists its GPU more than once: shards side by side);
    every cluster gets the result it gets alone, whichever shard it lands in, and the gather returns all of them."""
    from rpvg_amd import dist as rdist
    batch = synth.generate(seed=31, num_clusters=60, total_paths=2500, total_reads=60000)
    params = make_params(rng_seed=7)
    ref, _ = pyoracle.run(model, params, batch, 2)
    group = eng_mod.DeviceGroup(devices)
    try:
        got, secs = group.run(model, params, batch)
        assert secs > 0 and len(got) == batch.num_clusters
        for g, r in zip(got, ref):
            gk, rk = g.keyed(), r.keyed()
            assert set(gk) == set(rk)
            for key, (post, ab) in rk.items():
                assert small_cases.rel_close(gk[key][0], post, rel=1e-6) and small_cases.rel_close(gk[key][1], ab, rel=1e-6)
        where = group.partition()
        assert set(where.tolist()) == set(range(len(devices)))  # every shard has work
        costs = rdist.cluster_costs(batch)
        assert [sorted(np.nonzero(where == d)[0].tolist()) for d in range(len(devices))] == rdist.partition_clusters(costs, len(devices))
        flat = np.concatenate([g.abundances for g in got])
        gathered, tpm_den = group.gather(len(flat))
        assert np.array_equal(gathered, flat)
        assert abs(tpm_den - rdist.local_transcript_count(got, batch)) <= 1e-9 * tpm_den
        assert not group.has_communicator()  # one GPU: RCCL wants one rank per GPU
    finally:
        group.close()


@pytest.mark.parametrize("model", ["transcripts", "haplotype-transcripts", "haplotypes"])
def test_device_group_over_distinct_gpus_uses_its_communicator(model):
    """The first run of the RCCL collectives of a sharded batch — rpvg_hip_comm_init_all, rpvg_hip_gather
    (ncclAllGather, ragged), the TPM all-reduce — in a world larger than one: GPUs 0 and 1 of the box.  Skipped where
    there is one GPU (the round's test boxes): on a multi-GPU node the first multi-rank bench is then not also the
    first multi-rank RCCL call of this code.  `haplotypes` has no abundances: the gather of nothing must not wait."""
    from rpvg_amd import hip
    if hip.device_count() < 2:
        pytest.skip("one GPU: DeviceGroup over distinct GPUs needs two")
    batch = synth.generate(seed=33, num_clusters=80, total_paths=3000, total_reads=80000)
    params = make_params(rng_seed=7)
    ref, _ = pyoracle.run(model, params, batch, 2)
    group = eng_mod.DeviceGroup([0, 1])
    try:
        assert group.has_communicator()
        got, _ = group.run(model, params, batch)
        for g, r in zip(got, ref):
            gk, rk = g.keyed(), r.keyed()
            assert set(gk) == set(rk)
            for key, (post, ab) in rk.items():
                assert small_cases.rel_close(gk[key][0], post, rel=1e-6) and small_cases.rel_close(gk[key][1], ab, rel=1e-6)
        flat = np.concatenate([g.abundances for g in got]) if got else np.zeros(0)
        gathered, tpm_den = group.gather(len(flat))
        assert np.array_equal(gathered, flat)
        from rpvg_amd import dist as rdist
        assert abs(tpm_den - rdist.local_transcript_count(got, batch)) <= 1e-9 * max(1.0, tpm_den)
    finally:
        group.close()


def test_batches_arriving_through_an_uploader_engine():
    """Two resident slots, an uploader engine (rpvg_hip_create_uploader: a stream and hardware queue of its own) that
    re-uploads one slot from page-locked host arrays while the other is estimated: what bench.py's upload leg does.  The
    estimates of every batch equal those of a plain run, whichever engine copied its rows."""
    import threading
    from rpvg_amd import hip
    from tests import fuzz_parity
    batches = [ClusterBatch.from_clusters(small_cases.make_batch_clusters(1200 + i, n_clusters=12, with_empty=(i == 1))) for i in range(2)]
    params = make_params()
    engine = eng_mod.Engine(0)
    uploader = eng_mod.Engine(0, uploader=True)
    arrays = [a for b in batches for a in (b.cluster_row_off, b.cluster_path_off, b.row_count, b.row_noise, b.row_grp_off, b.grp_prob,
                                           b.grp_idx_off, b.path_idx)]
    for a in arrays:
        hip.host_register(a)
    try:
        want = [engine.run("haplotype-transcripts", params, engine.prepare(b))[0] for b in batches]
        slots = [engine.prepare(b) for b in batches]  # (a slot keeps the paths of its batch: the rows are what arrives)
        for k in range(4):
            other = slots[(k + 1) % 2]
            t = threading.Thread(target=other.reupload, args=(uploader,))
            t.start()
            got = engine.run("haplotype-transcripts", params, slots[k % 2])[0]
            t.join()
            assert not fuzz_parity.compare(got, want[k % 2])
    finally:
        for a in arrays:
            hip.host_unregister(a)
        uploader.close()
        engine.close()


@pytest.mark.parametrize("model,kw", [("haplotype-transcripts", {}), ("transcripts", {}), ("haplotypes", dict(use_hap_gibbs=1, rng_seed=5)),
                                      ("haplotype-transcripts", dict(num_gibbs_samples=3, rng_seed=9))],
                         ids=["nested", "transcripts", "haplotype-gibbs", "nested-read-count-samples"])
def test_batches_in_flight_equal_one_call_after_the_other(model, kw):
    """BatchPipeline (rpvg_amd/host/batch_pipeline.hpp): different batches in flight at once — an uploader thread, three
    estimator threads with an engine each — leave the estimates the same batches get from one engine, one call after the
    other, bit for bit (batches do not interact; cluster i of every batch draws from mt19937(rng_seed + i),
    src/main.cpp:976).  Also: containers handed in again (a slot reused by a later batch) hold that later batch's estimates,
    and a batch that fails (a path index outside its cluster) is reported by wait() without wedging the pipeline."""
    params = make_params(**kw)
    batches = [ClusterBatch.from_clusters(small_cases.make_batch_clusters(7300 + b, n_clusters=30, with_empty=True)) for b in range(3)]
    # (all batches of a set of containers have the same clusters: the same paths; the rows differ)
    base = small_cases.make_batch_clusters(7400, n_clusters=25, with_empty=True)
    rng = np.random.default_rng(7401)
    variants = []
    for v in range(5):
        clusters = []
        for cl in base:
            rows = [(int(c) + int(rng.integers(0, 3)), z, g) for (c, z, g) in cl["rows"]]
            clusters.append(dict(paths=cl["paths"], rows=rows))
        variants.append(ClusterBatch.from_clusters(clusters))
    engine = eng_mod.Engine(0)
    try:
        expected = [engine.run(model, params, engine.prepare(b))[0] for b in variants]
    finally:
        engine.close()
    pipe = eng_mod.Pipeline(model, params, 0, workers=3)
    try:
        pipe.prepare_slots(variants[0], 5)
        for round_ in range(2):  # the second round reuses every slot
            for v, b in enumerate(variants):
                pipe.submit(b, (v + round_) % 5, compact=(round_ == 1))  # (the second round with 32-bit offset arrays)
            pipe.wait()
            for v in range(len(variants)):
                got = pipe.result((v + round_) % 5)
                for k, (g, e) in enumerate(zip(got, expected[v])):
                    assert g.path_group_sets == e.path_group_sets, (v, k)
                    assert np.array_equal(g.posteriors, e.posteriors) and np.array_equal(g.abundances, e.abundances), (v, k)
                    assert g.noise_count == e.noise_count and g.total_count == e.total_count, (v, k)
                    assert g.em_iters == e.em_iters and g.em_cols == e.em_cols, (v, k)
                    assert len(g.gibbs_samples) == len(e.gibbs_samples), (v, k)
        # a bad batch between good ones
        bad_fields = {name: getattr(variants[1], name).copy() for name in ClusterBatch._DTYPES}
        bad_fields["path_idx"][0] = 10 ** 6
        pipe.submit(variants[0], 0)
        pipe.submit(ClusterBatch(**bad_fields), 1)
        with pytest.raises(eng_mod.hip.EngineError, match="refers to path"):
            pipe.wait()
        pipe.submit(variants[2], 2)
        pipe.wait()
        got = pipe.result(2)
        assert all(np.array_equal(g.posteriors, e.posteriors) for g, e in zip(got, expected[2]))
    finally:
        pipe.close()
    del batches


@pytest.mark.parametrize("model", ["haplotype-transcripts", "transcripts"])
def test_parts_of_one_data_set_equal_the_whole_batch(model):
    """One data set cut by cluster into parts (ClusterBatch.cluster_range: views of the caller's arrays, the long offset arrays as
    slices of its counts of one byte) that go through the pipeline one behind the other — what bench.py's `single_dataset_ms`
    times — leaves, cluster for cluster, the estimates of the whole batch in one call: bit for bit."""
    params = make_params()
    batch = synth.generate(seed=41, num_clusters=120, total_paths=2600, total_reads=200000)
    engine = eng_mod.Engine(0)
    try:
        whole, _ = engine.run(model, params, engine.prepare(batch))
    finally:
        engine.close()
    cuts = [0, 7, 30, 31, 80, batch.num_clusters]
    parts = [batch.cluster_range(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    assert sum(p.num_rows for p in parts) == batch.num_rows
    pipe = eng_mod.Pipeline(model, params, 0, workers=3)
    try:
        for slot, part in enumerate(parts):
            pipe.prepare_slot(slot, part)
        for _ in range(2):
            for slot, part in enumerate(parts):
                pipe.submit(part, slot, compact=True)
            pipe.wait()
            got = [e for slot in range(len(parts)) for e in pipe.result(slot)]
            assert len(got) == len(whole)
            for k, (g, e) in enumerate(zip(got, whole)):
                assert g.path_group_sets == e.path_group_sets, k
                assert np.array_equal(g.posteriors, e.posteriors) and np.array_equal(g.abundances, e.abundances), k
                assert g.noise_count == e.noise_count and g.total_count == e.total_count and g.em_iters == e.em_iters, k
    finally:
        pipe.close()
