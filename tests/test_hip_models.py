"""GPU parity of the three inference models, through the C++ host classes that keep the
reference's PathEstimator interface (rpvg_amd/host) and the C ABI under them, against the CPU oracle.

Both entry styles are exercised: estimateBatch() over a whole batch and the reference-shaped
per-cluster estimate() (src/path_estimator.hpp:23).  Results are compared keyed by group set, never by
row order (the reference's own order is that of a hash map: SURVEY.md H4).

Bar: group sets identical, totals and EM iteration counts exact, abundances/posteriors within 1e-4
relative (abs floor 1e-8).  Asserted at 1e-6: FP64 end to end leaves ~1e-10.
"""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle
from rpvg_amd import engine as eng_mod
from rpvg_amd.batch import ClusterBatch, make_params
from tests import small_cases

pytestmark = pytest.mark.gpu

REL = 1e-6
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def engine():
    e = eng_mod.Engine(0)
    yield e
    e.close()


def _compare(got, ref, check_iters=True, post_floor=1e-8):
    assert len(got) == len(ref)
    for k, (g, r) in enumerate(zip(got, ref)):
        gk, rk = g.keyed(), r.keyed()
        assert set(gk) == set(rk), f"cluster {k}: group sets differ"
        for key, (post, ab) in rk.items():
            assert small_cases.rel_close(gk[key][0], post, rel=REL, floor=post_floor), (k, key, gk[key][0], post)
            assert small_cases.rel_close(gk[key][1], ab, rel=REL), (k, key, gk[key][1], ab)
        assert g.total_count == r.total_count
        assert abs(g.noise_count - r.noise_count) <= REL * max(1.0, r.total_count)
        if check_iters:
            assert dict(zip(g.em_cols, g.em_iters)) == dict(zip(r.em_cols, r.em_iters)), f"cluster {k}: EM iterations differ"
        if r.total_count > 0 and len(r.abundances) and np.any(r.abundances):
            # invariant the reference's writers assert (src/threaded_output_writer.cpp:327-328)
            assert abs(g.abundances.sum() + g.noise_count - g.total_count) <= 1e-9 * g.total_count


def _golden(name):
    with open(os.path.join(GOLDEN, f"oracle_{name}.json")) as f:
        gold = json.load(f)
    clusters = [dict(paths=c["paths"], rows=[(r[0], r[1], [(g[0], g[1]) for g in r[2]]) for r in c["rows"]])
                for c in gold["clusters"]]
    return clusters, gold


@pytest.mark.parametrize("model", ["transcripts", "haplotype-transcripts", "haplotypes"])
def test_models_reproduce_golden_vectors(engine, model):
    clusters, gold = _golden(model)
    batch = ClusterBatch.from_clusters(clusters)
    prep = engine.prepare(batch)
    got, _ = engine.run(model, make_params(**gold["params"]), prep)
    for k, (g, ge) in enumerate(zip(got, gold["estimates"])):
        gk = g.keyed()
        want = {tuple(s[0]): (s[1], tuple(s[2])) for s in ge["sets"]}
        assert set(gk) == set(want), k
        for key, (post, ab) in want.items():
            assert small_cases.rel_close(gk[key][0], post, rel=REL)
            assert small_cases.rel_close(gk[key][1], ab, rel=REL)
        assert g.total_count == ge["total_count"]
        assert abs(g.noise_count - ge["noise_count"]) <= REL * max(1.0, ge["total_count"])
        assert sorted(g.em_iters) == sorted(ge["em_iters"])


@pytest.mark.parametrize("seed", [601, 602, 603])
@pytest.mark.parametrize("model", ["transcripts", "haplotype-transcripts", "haplotypes"])
def test_batch_matches_oracle(engine, model, seed):
    clusters = small_cases.make_batch_clusters(seed, n_clusters=16)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params()
    ref, _ = pyoracle.run(model, params, batch, 2)
    got, secs = engine.run(model, params, engine.prepare(batch))
    assert secs > 0
    _compare(got, ref)


@pytest.mark.parametrize("model", ["transcripts", "haplotype-transcripts", "haplotypes"])
def test_per_cluster_estimate_interface(engine, model):
    """The drop-in: PathEstimator::estimate(PathClusterEstimates*, vector<ReadPathProbabilities>&, mt19937*)."""
    clusters = small_cases.make_batch_clusters(611, n_clusters=8)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params()
    ref, _ = pyoracle.run(model, params, batch, 1)
    got, _ = engine.run(model, params, engine.prepare(batch, per_cluster=True))
    _compare(got, ref)


@pytest.mark.parametrize("ploidy", [1, 3])
def test_haplotypes_other_ploidies_use_full_enumeration(engine, ploidy):
    clusters = small_cases.make_batch_clusters(621, n_clusters=6)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params(ploidy=ploidy)
    ref, _ = pyoracle.run("haplotypes", params, batch, 1)
    got, _ = engine.run("haplotypes", params, engine.prepare(batch))
    for g, r in zip(got, ref):
        assert g.path_group_sets == r.path_group_sets  # Full: lexicographic multiset order is part of the contract
    _compare(got, ref)


def test_haplotype_transcripts_ploidy3(engine):
    clusters = small_cases.make_batch_clusters(631, n_clusters=5)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params(ploidy=3)
    ref, _ = pyoracle.run("haplotype-transcripts", params, batch, 1)
    got, _ = engine.run("haplotype-transcripts", params, engine.prepare(batch))
    _compare(got, ref)


def test_em_knobs_are_honoured(engine):
    clusters = small_cases.make_batch_clusters(641, n_clusters=6)
    batch = ClusterBatch.from_clusters(clusters)
    for kw in (dict(max_em_its=7), dict(max_rel_em_conv=1e-5), dict(min_hap_prob=0.05)):
        params = make_params(**kw)
        for model in ("transcripts", "haplotype-transcripts"):
            ref, _ = pyoracle.run(model, params, batch, 1)
            got, _ = engine.run(model, params, engine.prepare(batch))
            _compare(got, ref)


def test_unsupported_modes_fail_loudly(engine):
    from rpvg_amd import hip
    clusters = small_cases.make_batch_clusters(651, n_clusters=2, with_empty=False)
    prep = engine.prepare(ClusterBatch.from_clusters(clusters))
    with pytest.raises(hip.EngineError):
        engine.run("no-such-model", make_params(), prep)


def test_larger_cluster_bounded_search_matches_sequential_reference(engine):
    """More haplotype columns than the first fetch block: exercises the multi-round branch-and-bound."""
    rng = np.random.default_rng(661)
    clusters = [small_cases.make_cluster(rng, 3, [9, 7, 8], n_haps=40, n_reads=1500),
                small_cases.make_cluster(rng, 1, [30], n_haps=60, n_reads=800)]
    batch = ClusterBatch.from_clusters(clusters)
    for model in ("haplotype-transcripts", "haplotypes"):
        ref, _ = pyoracle.run(model, make_params(), batch, 2)
        got, _ = engine.run(model, make_params(), engine.prepare(batch))
        # posteriors below prob_precision are not reported by the reference's writers
        # (src/threaded_output_writer.cpp:260,473); compare them with an absolute floor
        _compare(got, ref)


def test_host_driven_bounded_search_agrees_with_on_device_search(engine):
    """Two implementations of the same sequential branch-and-bound (device kernel vs host replay of
    GPU-computed pair log-likelihoods) must keep the same pairs."""
    rng = np.random.default_rng(671)
    clusters = small_cases.make_batch_clusters(672, n_clusters=10)
    clusters.append(small_cases.make_cluster(rng, 2, [12, 9], n_haps=50, n_reads=2000))
    batch = ClusterBatch.from_clusters(clusters)
    for model in ("haplotype-transcripts", "haplotypes"):
        dev, _ = engine.run(model, make_params(), engine.prepare(batch))
        os.environ["RPVG_AMD_HOST_BOUNDED"] = "1"
        try:
            host, _ = engine.run(model, make_params(), engine.prepare(batch))
        finally:
            del os.environ["RPVG_AMD_HOST_BOUNDED"]
        for d, h in zip(dev, host):
            assert d.path_group_sets == h.path_group_sets
        _compare(dev, host)


def test_independent_haplotype_inference_matches_oracle(engine):
    """--ind-hap-inference (src/path_abundance_estimator.cpp:356-426): per-transcript posteriors on the GPU,
    subset sampling with the reference's mt19937(rng_seed + i) / discrete_distribution on the host."""
    clusters = small_cases.make_batch_clusters(681, n_clusters=10)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params(ind_hap_inference=1, rng_seed=7)
    ref, _ = pyoracle.run("haplotype-transcripts", params, batch, 1)
    got, _ = engine.run("haplotype-transcripts", params, engine.prepare(batch))
    _compare(got, ref)
    # the drop-in estimate() consumes the caller's generator the same way
    got1, _ = engine.run("haplotype-transcripts", params, engine.prepare(batch, per_cluster=True))
    _compare(got1, ref)


@pytest.mark.parametrize("model", ["haplotypes", "haplotype-transcripts"])
@pytest.mark.parametrize("ploidy", [1, 2])
def test_gibbs_haplotype_posteriors_follow_the_reference_stream(engine, model, ploidy):
    """--use-hap-gibbs (src/path_estimator.cpp:475-589): chains on the host with the reference's mt19937 and
    discrete_distribution, conditionals on the GPU.  Same libstdc++ streams as the oracle, so the sampled
    group sets and their frequencies agree draw for draw (a conditional differing in the last bits could
    flip a draw; none does on these inputs)."""
    clusters = small_cases.make_batch_clusters(691, n_clusters=8)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params(use_hap_gibbs=1, ploidy=ploidy, rng_seed=11)
    ref, _ = pyoracle.run(model, params, batch, 1)
    got, _ = engine.run(model, params, engine.prepare(batch))
    _compare(got, ref)
    for g, r in zip(got, ref):
        if model == "haplotypes":
            assert g.path_group_sets == r.path_group_sets  # first-seen order of the sampled sets
            assert abs(g.posteriors.sum() - (1.0 if r.path_group_sets else 0.0)) < 1e-12
    got1, _ = engine.run(model, params, engine.prepare(batch, per_cluster=True))
    _compare(got1, ref)


def test_gibbs_sampler_without_room_on_the_device_falls_back_to_the_host_driven_one(engine):
    """rpvg_hip_group_gibbs reports RPVG_HIP_ERR_UNSUPPORTED when the conditional distributions outgrow the memory set
    aside for them; the host-driven sampler (chains on the host, conditionals from rpvg_hip_group_conditionals) then runs
    on generators nobody has moved: the same draws, the same estimates; and ploidy 3 always takes that path."""
    clusters = small_cases.make_batch_clusters(691, n_clusters=8)
    batch = ClusterBatch.from_clusters(clusters)
    for model, ploidy in (("haplotypes", 2), ("haplotype-transcripts", 2), ("haplotypes", 3)):
        params = make_params(use_hap_gibbs=1, ploidy=ploidy, rng_seed=11)
        ref, _ = pyoracle.run(model, params, batch, 1)
        os.environ["RPVG_HIP_GIBBS_BYTES"] = "64"
        try:
            got, _ = engine.run(model, params, engine.prepare(batch))
        finally:
            del os.environ["RPVG_HIP_GIBBS_BYTES"]
        _compare(got, ref)
        again, _ = engine.run(model, params, engine.prepare(batch))
        _compare(again, ref)


@pytest.mark.parametrize("seed", [40061, 40136, 40173, 40005, 40118])
def test_gibbs_seeds_of_the_parity_sweep(engine, seed):
    """Fixed seeds of `tests/fuzz_parity.py <rounds> 40000 gibbs` (every case a haplotype model with --use-hap-gibbs).  The
    first three caught the device sampler reading its first generator words from a window in LDS it had not filled yet (the
    first chain of a call's first generator: one subset sample in eleven came out differently in the lanes' first clusters);
    the host-driven sampler agreed with the oracle on them."""
    from tests import fuzz_parity
    case = fuzz_parity.draw_case(seed, only_gibbs=True)
    assert case["kw"]["use_hap_gibbs"] == 1
    assert fuzz_parity.run_case(engine, case, oracle_threads=8) == []


@pytest.mark.parametrize("ploidy,seed", [(2, 693), (1, 694), (3, 695)])
def test_independent_inference_with_gibbs_posteriors_follows_the_reference_draw_for_draw(engine, ploidy, seed):
    """--ind-hap-inference with --use-hap-gibbs: the reference runs a transcript's chains and samples that transcript's subsets
    from the same generator before the next transcript's chains (src/path_abundance_estimator.cpp:371-412).  The batch runs
    transcript j of every cluster in round j and draws its subsets before round j + 1, so the generator is consumed in the
    reference's order: the subset samples, and everything after them, are the oracle's."""
    from tests import fuzz_parity
    clusters = small_cases.make_batch_clusters(seed, n_clusters=10, max_reads=300, with_empty=True)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params(use_hap_gibbs=1, ind_hap_inference=1, ploidy=ploidy, rng_seed=5)
    ref, _ = pyoracle.run("haplotype-transcripts", params, batch, 1)
    got, _ = engine.run("haplotype-transcripts", params, engine.prepare(batch))
    assert fuzz_parity.compare(got, ref) == []


def test_gibbs_posteriors_agree_with_exact_posteriors_statistically(engine):
    """Size-independent property: the Gibbs frequencies of the dominant diplotypes approach the exact
    (branch-and-bound) posteriors."""
    rng = np.random.default_rng(693)
    clusters = [small_cases.make_cluster(rng, 1, [4], n_haps=6, n_reads=40) for _ in range(6)]
    batch = ClusterBatch.from_clusters(clusters)
    exact, _ = engine.run("haplotypes", make_params(), engine.prepare(batch))
    gibbs, _ = engine.run("haplotypes", make_params(use_hap_gibbs=1, rng_seed=3), engine.prepare(batch))
    for e, g in zip(exact, gibbs):
        ek = {tuple(sorted(k)): v[0] for k, v in e.keyed().items()}
        gk = {tuple(sorted(k)): v[0] for k, v in g.keyed().items()}
        for key, p in ek.items():
            if p > 0.2:
                assert abs(gk.get(key, 0.0) - p) < 0.08, (key, p, gk.get(key))


# ---- -n > 0: Gibbs read-count samples (src/path_abundance_estimator.cpp:116-212) ---------------------
# Parity is statistical: the reference draws from mt19937 through libstdc++ distributions, the GPU from a
# counter-based Philox generator.

def _sample_stats(est):
    out = []
    for e in est:
        per = []
        for ids, noise, ab in e.gibbs_samples:
            per.append((ids, ab.mean(axis=0), ab.std(axis=0, ddof=1) if len(noise) > 1 else np.zeros(len(ids)), noise, ab))
        out.append(per)
    return out


def test_gibbs_read_count_samples_transcripts(engine):
    clusters = small_cases.make_batch_clusters(801, n_clusters=6, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    n, thin = 400, 5
    params = make_params(num_gibbs_samples=n, gibbs_thin_its=thin, rng_seed=5)
    ref, _ = pyoracle.run("transcripts", params, batch, 2)
    got, _ = engine.run("transcripts", params, engine.prepare(batch))
    _compare(got, ref)  # the EM part is unchanged by -n
    for k, (g, r) in enumerate(zip(got, ref)):
        assert len(g.gibbs_samples) == len(r.gibbs_samples) == 1
        (gi, gn, ga), (ri, rn, ra) = g.gibbs_samples[0], r.gibbs_samples[0]
        assert gi == ri == tuple(range(len(clusters[k]["paths"])))
        assert ga.shape == ra.shape == (n, len(gi)) and gn.shape == rn.shape == (n,)
        # every recorded state conserves the read mass (writers assert this on the EM result)
        assert np.all(np.abs(ga.sum(axis=1) + gn - g.total_count) <= 1e-9 * g.total_count)
        assert np.all(ga >= 0) and np.all(gn >= 0)
        # means agree within Monte-Carlo error (5 sigma of the difference of two means, + a floor)
        se = np.sqrt(ga.var(axis=0, ddof=1) / n + ra.var(axis=0, ddof=1) / n)
        assert np.all(np.abs(ga.mean(axis=0) - ra.mean(axis=0)) <= 5 * se + 0.02 * g.total_count / max(1, len(gi)) + 0.5), k
        # spreads agree within a factor
        sg, sr = ga.std(axis=0, ddof=1), ra.std(axis=0, ddof=1)
        big = sr > 1.0
        assert np.all(sg[big] < 1.6 * sr[big]) and np.all(sg[big] > 0.6 * sr[big])
        # and the samples scatter around the EM estimate
        assert np.all(np.abs(ga.mean(axis=0) - g.abundances) <= 6 * sg / np.sqrt(n) + 0.05 * g.total_count + 1.0)


def test_gibbs_read_count_samples_are_reproducible_and_seeded(engine):
    clusters = small_cases.make_batch_clusters(811, n_clusters=3, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    a, _ = engine.run("transcripts", make_params(num_gibbs_samples=20, gibbs_thin_its=3, rng_seed=1), engine.prepare(batch))
    b, _ = engine.run("transcripts", make_params(num_gibbs_samples=20, gibbs_thin_its=3, rng_seed=1), engine.prepare(batch))
    c, _ = engine.run("transcripts", make_params(num_gibbs_samples=20, gibbs_thin_its=3, rng_seed=2), engine.prepare(batch))
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x.gibbs_samples[0][2], y.gibbs_samples[0][2])
        assert not np.array_equal(x.gibbs_samples[0][2], z.gibbs_samples[0][2])


def test_gibbs_read_count_samples_nested(engine):
    clusters = small_cases.make_batch_clusters(821, n_clusters=6, with_empty=False)
    batch = ClusterBatch.from_clusters(clusters)
    n = 60
    params = make_params(num_gibbs_samples=n, gibbs_thin_its=4, rng_seed=9)
    ref, _ = pyoracle.run("haplotype-transcripts", params, batch, 2)
    got, _ = engine.run("haplotype-transcripts", params, engine.prepare(batch))
    _compare(got, ref)
    for g, r in zip(got, ref):
        # the -n samples of a cluster are split over its path subsets (binomially on the subset weights)
        assert sum(len(s[1]) for s in g.gibbs_samples) == sum(len(s[1]) for s in r.gibbs_samples) <= n
        subsets = set(g.em_cols)
        for ids, noise, ab in g.gibbs_samples:
            assert ids in subsets
            assert ab.shape == (len(noise), len(ids))
            assert np.all(np.abs(ab.sum(axis=1) + noise - g.total_count) <= 1e-9 * g.total_count)


# ---- -i strains: minimum path cover (src/path_abundance_estimator.cpp:217-340) -------------------------

def test_min_path_cover_reference_case_on_device(hip_ctx):
    """The reference's own test (src/tests/path_abundance_estimator_test.cpp:8-28): cover rows
    [1,0,1],[0,1,0],[1,0,0],[0,1,1], counts [1,3,1,5]; weights [1,1,1] -> {0,1}; weight_2 = 0.01 -> {0,1,2}.
    The device entry derives the weights from the rows (-sum count*log(prob)), so the probabilities are
    chosen to give exactly those weights."""
    import math

    def cluster(w2):
        x = {0: 1 / 2.0, 1: 1 / 8.0, 2: w2 / 6.0}  # sum over a path's rows of count * x = weight
        rows = [(1, 1e-4, [(math.exp(-x[0]), [0]), (math.exp(-x[2]), [2])]),
                (3, 1e-4, [(math.exp(-x[1]), [1])]),
                (1, 1e-4, [(math.exp(-x[0]), [0])]),
                (5, 1e-4, [(math.exp(-x[1]), [1]), (math.exp(-x[2]), [2])])]
        rows = [(c, n, sorted(g)) for c, n, g in rows]
        return dict(paths=[{}, {}, {}], rows=rows)

    batch = ClusterBatch.from_clusters([cluster(1.0), cluster(0.01), dict(paths=[{}], rows=[(2, 0.1, [(0.9, [0])])])])
    dev = hip_ctx.upload(batch)
    assert hip_ctx.min_path_cover(dev, [0, 1, 2]) == [[0, 1], [0, 1, 2], [0]]


@pytest.mark.parametrize("seed", [901, 902])
def test_strains_model_matches_oracle(engine, seed):
    clusters = small_cases.make_batch_clusters(seed, n_clusters=14)
    batch = ClusterBatch.from_clusters(clusters)
    ref, _ = pyoracle.run("strains", make_params(), batch, 2)
    got, _ = engine.run("strains", make_params(), engine.prepare(batch))
    for g, r in zip(got, ref):
        assert g.em_cols == r.em_cols  # the cover itself: integer result, exact
    _compare(got, ref)
    got1, _ = engine.run("strains", make_params(), engine.prepare(batch, per_cluster=True))
    _compare(got1, ref)


# ---- host lanes (rpvg_amd/host/pipeline_lanes.hpp) ------------------------------------------------------------------

def test_batches_large_enough_for_two_lanes_match_oracle(engine):
    """Batches of 64+ clusters are cut in two host lanes (second lane: own thread and device context); every
    cluster must come out exactly as in the oracle's one-by-one loop, empty clusters included."""
    clusters = small_cases.make_batch_clusters(681, n_clusters=150, max_reads=120)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params()
    ref, _ = pyoracle.run("haplotype-transcripts", params, batch, 2)
    prep = engine.prepare(batch)
    for _ in range(3):  # the lanes are reused from call to call
        got, _ = engine.run("haplotype-transcripts", params, prep)
        _compare(got, ref)


def test_an_error_in_the_second_lane_reaches_the_caller(engine):
    from rpvg_amd import hip
    clusters = small_cases.make_batch_clusters(682, n_clusters=100, max_reads=60, with_empty=False)
    for p in clusters[71]["paths"]:  # an odd position: the second lane's share
        p["source_ids"] = []
    prep = engine.prepare(ClusterBatch.from_clusters(clusters))
    with pytest.raises(hip.EngineError, match="source"):
        engine.run("haplotype-transcripts", make_params(), prep)
    # the engine and its lanes stay usable
    good = small_cases.make_batch_clusters(683, n_clusters=80, max_reads=60)
    batch = ClusterBatch.from_clusters(good)
    ref, _ = pyoracle.run("haplotype-transcripts", make_params(), batch, 2)
    got, _ = engine.run("haplotype-transcripts", make_params(), engine.prepare(batch))
    _compare(got, ref)


def test_lane_count_does_not_change_the_result():
    """RPVG_AMD_LANES is read once per process: compare 1, 2 and 3 lanes in child processes."""
    import subprocess
    import sys
    code = (
        "import numpy as np, zlib\n"
        "from rpvg_amd import engine as e, synth\n"
        "from rpvg_amd.batch import make_params\n"
        "b = synth.generate(seed=7, num_clusters=300, total_paths=9000, total_reads=300000)\n"
        "eng = e.Engine(0)\n"
        "est, _ = eng.run('haplotype-transcripts', make_params(), eng.prepare(b))\n"
        "sets = sum(len(x.path_group_sets) for x in est)\n"
        "its = sum(int(np.sum(x.em_iters)) for x in est)\n"
        "post = np.concatenate([x.posteriors for x in est])\n"
        "ab = np.concatenate([x.abundances for x in est])\n"
        "print(sets, its, zlib.crc32(post.tobytes()), round(float(ab.sum()), 3))\n")
    outs = []
    for lanes in ("1", "2", "3"):
        env = dict(os.environ, RPVG_AMD_LANES=lanes)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(res.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2], outs


@pytest.mark.parametrize("model", ["transcripts", "strains", "haplotype-transcripts", "haplotypes"])
def test_reference_shaped_factory_runs_on_the_default_engine(model):
    """The reference's factory block (src/main.cpp:766-788) and per-cluster estimate() call (:976-977) compiled against the
    host library: two disjoint paths with 30 and 70 reads (KAT-EM-disjoint)."""
    import subprocess
    binary = small_cases.build_reference_factory()
    out = subprocess.run([binary, model], capture_output=True, text=True, check=True).stdout.split()
    assert out[0] == model
    if model != "haplotypes":
        assert float(out[out.index("total") + 1]) == 100.0
        ab = [float(x) for x in out[out.index("abundances") + 1:out.index("posteriors")]]
        assert abs(sum(ab) - 100.0) < 1e-6 and sorted(round(a, 6) for a in ab if a > 0) == [30.0, 70.0]
    else:
        post = [float(x) for x in out[out.index("posteriors") + 1:]]
        assert abs(sum(post) - 1.0) < 1e-9


def _flat_posterior_clusters(seed, n=3):
    """Clusters with many haplotype columns and a handful of reads: hundreds of diplotypes pass min_hap_prob."""
    rng = np.random.default_rng(seed)
    return ([small_cases.make_cluster(rng, 3, (12, 10, 8), n_haps=30, n_reads=int(rng.integers(3, 7)), empty_read_frac=0.0) for _ in range(n)] +
            [small_cases.make_cluster(rng, 1, (36,), n_haps=36, n_reads=2, empty_read_frac=0.0),
             small_cases.make_cluster(rng, 2, (20, 18), n_haps=30, n_reads=1, empty_read_frac=0.0)])


def test_device_subset_path_equals_the_separate_calls(engine, monkeypatch):
    """`-i haplotype-transcripts` through rpvg_hip_nested_subset_em (search -> subsets -> EM on the device) against the
    three separate calls with the host in between (RPVG_HIP_NO_DEVICE_SUBSETS=1): the same path subsets, weights equal to
    the bit (the same additions in the same order: src/path_abundance_estimator.cpp:595-605), the same EM iteration
    counts; both against the oracle.  The flat-posterior clusters take the select kernel's wide variant (more than 64 kept
    diplotypes per matrix) and give a single matrix hundreds of retained subsets."""
    clusters = small_cases.make_batch_clusters(8801, n_clusters=40, with_empty=True) + _flat_posterior_clusters(8802)
    batch = ClusterBatch.from_clusters(clusters)
    prep = engine.prepare(batch)
    params = make_params()
    fused, _ = engine.run("haplotype-transcripts", params, prep)
    monkeypatch.setenv("RPVG_HIP_NO_DEVICE_SUBSETS", "1")
    separate, _ = engine.run("haplotype-transcripts", params, prep)
    monkeypatch.delenv("RPVG_HIP_NO_DEVICE_SUBSETS")
    most_subsets = 0
    for k, (f, s) in enumerate(zip(fused, separate)):
        fk, sk = f.keyed(), s.keyed()
        assert set(fk) == set(sk), k
        for key in sk:
            assert fk[key][0] == sk[key][0], (k, key)  # posteriors: sums of subset weights in the same order
            assert small_cases.rel_close(fk[key][1], sk[key][1], rel=1e-12, floor=1e-300), (k, key)
        assert list(f.em_cols) == list(s.em_cols) and list(f.em_iters) == list(s.em_iters), k
        assert f.total_count == s.total_count and abs(f.noise_count - s.noise_count) <= 1e-12 * max(1.0, s.total_count)
        most_subsets = max(most_subsets, len(f.em_iters))
    assert most_subsets > 64, most_subsets  # the wide select kernel was exercised
    ref, _ = pyoracle.run("haplotype-transcripts", params, batch, 4)
    _compare(fused, ref)


@pytest.mark.parametrize("model,kw", [("haplotype-transcripts", {}), ("transcripts", {}), ("haplotypes", dict(use_hap_gibbs=1, rng_seed=21)),
                                      ("haplotype-transcripts", dict(ind_hap_inference=1, rng_seed=4)), ("strains", {})],
                         ids=["nested", "transcripts", "haplotype-gibbs", "nested-independent", "strains"])
@pytest.mark.parametrize("threads", [64, 5])
def test_team_of_threads_calling_estimate_equals_the_batch(engine, model, kw, threads):
    """The reference's cluster loop (src/main.cpp:829,976-977): estimate() once per cluster from every thread of an OpenMP
    team, cluster i with mt19937(rng_seed + i).  The calls in flight are joined into batches behind the interface
    (PathEstimator::CallCombiner) — which clusters share a batch depends on the scheduler, the estimates of a cluster must
    not: they equal those of estimateBatch() on all clusters at once, and the oracle's."""
    clusters = small_cases.make_batch_clusters(6400, n_clusters=150, with_empty=True)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params(**kw)
    whole, _ = engine.run(model, params, engine.prepare(batch))
    for repeat in range(2):  # (different batch compositions)
        team, secs = engine.run_team(model, params, engine.prepare(batch, per_cluster=True), threads)
        assert secs > 0
        for k, (t, w) in enumerate(zip(team, whole)):
            assert t.path_group_sets == w.path_group_sets, k
            assert np.allclose(t.posteriors, w.posteriors, rtol=1e-12, atol=0) and np.allclose(t.abundances, w.abundances, rtol=1e-9, atol=1e-12), k
            assert abs(t.noise_count - w.noise_count) <= 1e-9 * max(1.0, w.total_count) and t.total_count == w.total_count, k
            assert t.em_iters == w.em_iters and t.em_cols == w.em_cols, k
    if not kw:
        ref, _ = pyoracle.run(model, params, batch, 4)
        _compare(team, ref)


def test_a_team_larger_than_a_batch_may_be(engine):
    """More callers than the combiner joins into one batch (256) and many more than the host has cores — the configuration that
    serves estimate() best, its callers being asleep while their batch runs (profiles/r06/a1_by_team_size.txt): 320 threads over
    700 clusters, the estimates of every cluster those of estimateBatch() on all of them."""
    clusters = small_cases.make_batch_clusters(6500, n_clusters=700, with_empty=True)
    batch = ClusterBatch.from_clusters(clusters)
    params = make_params()
    whole, _ = engine.run("haplotype-transcripts", params, engine.prepare(batch))
    team, _ = engine.run_team("haplotype-transcripts", params, engine.prepare(batch, per_cluster=True), 320)
    for k, (t, w) in enumerate(zip(team, whole)):
        assert t.path_group_sets == w.path_group_sets, k
        assert np.array_equal(t.posteriors, w.posteriors) and np.array_equal(t.abundances, w.abundances), k
        assert t.noise_count == w.noise_count and t.total_count == w.total_count and t.em_iters == w.em_iters, k
