"""File formats either side of the hot path, CPU only:
`--write-probs` dump reader/writer (src/threaded_output_writer.cpp:42-95), `-f` path info parser
(src/main.cpp:239-353) and the result TSV writers (src/threaded_output_writer.cpp:98-546)."""
import gzip
import os

import numpy as np
import pytest

from oracle import pyoracle
from rpvg_amd import io as rio, synth
from rpvg_amd.batch import ClusterBatch, make_params
from tests import small_cases


def fmt(x):
    """std::setprecision(8) with default float formatting."""
    return "%.8g" % x


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("io")
    batch = synth.generate(seed=31, num_clusters=25, total_paths=500, total_reads=20000)
    probs, info = str(d / "run_probs.txt.gz"), str(d / "info.tsv")
    rio.write_batch_files(batch, probs, info)
    return dict(dir=d, batch=batch, probs=probs, info=info)


def test_probs_dump_format_and_round_trip(files):
    text = gzip.open(files["probs"], "rt").read().splitlines()
    assert text[0] == "#"
    first_paths = text[1].split(" ")
    name, length, eff = first_paths[0].rsplit(",", 2)
    assert name == "c0_p0" and int(length) > 0 and float(eff) > 0
    cnt, noise, *groups = text[2].split(" ")
    assert int(cnt) >= 1 and 0 < float(noise) <= 1
    for g in groups:
        prob, idx = g.split(":")
        assert float(prob) > 0 and all(int(i) < len(first_paths) for i in idx.split(","))
    assert sum(1 for line in text if line == "#") == files["batch"].num_clusters

    back = rio.read_batch_files(files["probs"], files["info"])
    orig = files["batch"]
    assert back.num_clusters == orig.num_clusters and back.num_rows == orig.num_rows and back.num_paths == orig.num_paths
    assert back.total_reads == orig.total_reads
    # the generator already emits clusters by descending read count, so the ranking keeps the order
    assert np.array_equal(back.cluster_row_off, orig.cluster_row_off)
    assert np.array_equal(back.row_count, orig.row_count)
    assert np.array_equal(back.path_idx, orig.path_idx)
    # transcript ids are re-numbered densely over the whole file (first seen); per cluster the partition is the same
    for k in range(orig.num_clusters):
        p0, p1 = int(orig.cluster_path_off[k]), int(orig.cluster_path_off[k + 1])
        a, b = orig.path_group_id[p0:p1], back.path_group_id[p0:p1]
        assert len(set(zip(a.tolist(), b.tolist()))) == len(set(a.tolist())) == len(set(b.tolist()))
    # values carry 8 significant digits in the dump
    assert np.allclose(back.grp_prob, orig.grp_prob, rtol=1e-7, atol=0)
    assert np.allclose(back.row_noise, orig.row_noise, rtol=1e-7, atol=0)
    # haplotype ids are re-numbered densely in first-seen order; the grouping they induce is unchanged
    for k in (0, 7, 24):
        a, b = orig.cluster(k)["paths"], back.cluster(k)["paths"]
        rel = {}
        for pa, pb in zip(a, b):
            assert len(pa["source_ids"]) == len(pb["source_ids"]) == pb["source_count"]
        by_a = {s: tuple(i for i, p in enumerate(a) if s in p["source_ids"]) for p in a for s in p["source_ids"]}
        by_b = {s: tuple(i for i, p in enumerate(b) if s in p["source_ids"]) for p in b for s in p["source_ids"]}
        assert sorted(by_a.values()) == sorted(by_b.values())


def test_path_info_parser_formats(tmp_path):
    new = tmp_path / "new.tsv"
    new.write_text("Name\tLength\tTranscript\tHaplotypes\nh1\t100\ttA\tx,y\nh2\t90\ttA\ty\nh3\t50\ttB\tz,x,w\n")
    old = tmp_path / "old.tsv"
    old.write_text("Name\tLength\tTranscript\tReference\tHaplotypes\nh1\t100\ttA\tref1\tx,y\nh2\t90\ttA\tref1\ty\nh3\t50\ttB\tref2\tz,x,w\n")
    probs = tmp_path / "p.txt"
    probs.write_text("#\nh1,100,80 h2,90,70 h3,50,30\n3 0.001 0.3:0,1 0.5:2\n")
    for info in (new, old):
        b = rio.read_batch_files(str(probs), str(info), parse_haplotype_ids=True)
        cl = b.cluster(0)
        assert [p["group_id"] for p in cl["paths"]] == [0, 0, 1]          # dense transcript ids, first seen
        assert [p["source_ids"] for p in cl["paths"]] == [[0, 1], [1], [0, 2, 3]]  # dense haplotype ids, first seen
        assert [p["source_count"] for p in cl["paths"]] == [2, 1, 3]
        assert cl["rows"] == [(3, 0.001, [(0.3, [0, 1]), (0.5, [2])])]
        b2 = rio.read_batch_files(str(probs), str(info), parse_haplotype_ids=False)
        assert [p["source_count"] for p in b2.cluster(0)["paths"]] == [2, 1, 3] and not b2.source_id.size
    dup = tmp_path / "dup.tsv"
    dup.write_text("Name\tLength\tTranscript\tHaplotypes\nh1\t1\tt\tx\nh1\t1\tt\ty\n")
    from rpvg_amd import hip
    with pytest.raises(hip.EngineError):
        rio.read_batch_files(str(probs), str(dup))
    missing = tmp_path / "missing.tsv"
    missing.write_text("Name\tLength\tTranscript\tHaplotypes\nh1\t1\tt\tx\n")
    with pytest.raises(hip.EngineError):
        rio.read_batch_files(str(probs), str(missing))


def _names(batch, k):
    return [f"c{k}_p{j}" for j in range(int(batch.cluster_path_off[k + 1] - batch.cluster_path_off[k]))]


def test_abundance_writer_format(files):
    batch = rio.read_batch_files(files["probs"], files["info"])
    params = make_params()
    prefix = str(files["dir"] / "tx")
    with pyoracle.RawRun("transcripts", params, batch, 2) as run:
        rio.write_estimates(files["probs"], files["info"], "transcripts", params, run.view, prefix, unaligned_read_count=17)
        est = run.estimates
    lines = open(prefix + ".txt").read().splitlines()
    assert lines[0] == "Name\tClusterID\tLength\tEffectiveLength\tReadCount\tTPM"
    eff = batch.path_effective_length
    total_tc = sum(a / eff[int(batch.cluster_path_off[k]) + j] for k, e in enumerate(est) for j, a in enumerate(e.abundances))
    row = 1
    for k, e in enumerate(est):
        for j, a in enumerate(e.abundances):
            el = eff[int(batch.cluster_path_off[k]) + j]
            want = [f"c{k}_p{j}", str(k + 1), str(int(el) + 50), fmt(el), fmt(a), fmt(a / el / total_tc * 1e6)]
            assert lines[row].split("\t") == want, (row, lines[row], want)
            row += 1
    assert lines[row] == "Unknown\t0\t0\t0\t" + fmt(sum(e.noise_count for e in est) + 17) + "\t0"
    assert row + 1 == len(lines)
    tpm = sum(float(l.split("\t")[5]) for l in lines[1:-1])
    assert abs(tpm - 1e6) < 1.0


def test_haplotype_transcript_writers_format(files):
    batch = rio.read_batch_files(files["probs"], files["info"])
    params = make_params()
    prefix = str(files["dir"] / "ht")
    with pyoracle.RawRun("haplotype-transcripts", params, batch, 2) as run:
        rio.write_estimates(files["probs"], files["info"], "haplotype-transcripts", params, run.view, prefix, unaligned_read_count=4)
        est = run.estimates
    eff = batch.path_effective_length
    total_tc = 0.0
    for k, e in enumerate(est):
        a = 0
        for s in e.path_group_sets:
            for p in s:
                total_tc += e.abundances[a] / eff[int(batch.cluster_path_off[k]) + p]
                a += 1
    # <prefix>.txt: one row per path
    lines = open(prefix + ".txt").read().splitlines()
    assert lines[0] == "Name\tClusterID\tLength\tEffectiveLength\tHaplotypeProbability\tReadCount\tTPM"
    row = 1
    for k, e in enumerate(est):
        n = int(batch.cluster_path_off[k + 1] - batch.cluster_path_off[k])
        prob, cnt = np.zeros(n), np.zeros(n)
        a = 0
        for s, post in zip(e.path_group_sets, e.posteriors):
            for j, p in enumerate(s):
                if j == 0 or s[j] != s[j - 1]:
                    prob[p] += post
                cnt[p] += e.abundances[a]
                a += 1
        for j in range(n):
            el = eff[int(batch.cluster_path_off[k]) + j]
            want = [f"c{k}_p{j}", str(k + 1), str(int(el) + 50), fmt(el), fmt(prob[j]), fmt(cnt[j]), fmt(cnt[j] / el / total_tc * 1e6)]
            assert lines[row].split("\t") == want, (row, lines[row], want)
            row += 1
    assert lines[row] == "Unknown\t0\t0\t0\t0\t" + fmt(sum(e.noise_count for e in est) + 4) + "\t0"
    # <prefix>_joint.txt: one row per group set
    joint = open(prefix + "_joint.txt").read().splitlines()
    assert joint[0] == "Name_1\tName_2\tClusterID\tHaplotypingProbability\tReadCount_1\tTPM_1\tReadCount_2\tTPM_2"
    row = 1
    for k, e in enumerate(est):
        a = 0
        for s, post in zip(e.path_group_sets, e.posteriors):
            names = [f"c{k}_p{p}" for p in s] + ["."] * (2 - len(s))
            vals = []
            for p in s:
                el = eff[int(batch.cluster_path_off[k]) + p]
                vals += [fmt(e.abundances[a]), fmt(e.abundances[a] / el / total_tc * 1e6)]
                a += 1
            vals += ["0", "0"] * (2 - len(s))
            assert joint[row].split("\t") == names + [str(k + 1), fmt(post)] + vals, (row, joint[row])
            row += 1
    half = sum(e.noise_count / 2 for e in est) + 4 / 2
    assert joint[row] == "Unknown\tUnknown\t0\t0\t" + fmt(half) + "\t0\t" + fmt(half) + "\t0"


def test_haplotypes_and_gibbs_writers_format(files):
    batch = rio.read_batch_files(files["probs"], files["info"], parse_haplotype_ids=False)
    params = make_params()
    prefix = str(files["dir"] / "hap")
    with pyoracle.RawRun("haplotypes", params, batch, 2) as run:
        rio.write_estimates(files["probs"], files["info"], "haplotypes", params, run.view, prefix)
        est = run.estimates
    lines = open(prefix + ".txt").read().splitlines()
    assert lines[0] == "Name_1\tName_2\tClusterID\tHaplotypingProbability"
    want = [[f"c{k}_p{s[0]}", f"c{k}_p{s[1]}", str(k + 1), fmt(p)] for k, e in enumerate(est)
            for s, p in zip(e.path_group_sets, e.posteriors) if p >= 1e-8]
    assert [l.split("\t") for l in lines[1:]] == want

    n = 6
    params = make_params(num_gibbs_samples=n, gibbs_thin_its=2, rng_seed=3)
    prefix = str(files["dir"] / "g")
    with pyoracle.RawRun("transcripts", params, batch, 2) as run:
        rio.write_estimates(files["probs"], files["info"], "transcripts", params, run.view, prefix, unaligned_read_count=5)
        est = run.estimates
    lines = gzip.open(prefix + "_gibbs.txt.gz", "rt").read().splitlines()
    assert lines[0] == "Name\tClusterID" + "".join(f"\tReadCountSample_{i + 1}" for i in range(n))
    row = 1
    noise = np.zeros(n)
    for k, e in enumerate(est):
        (ids, ns, ab), = e.gibbs_samples
        noise += ns
        for j, p in enumerate(ids):
            assert lines[row].split("\t") == [f"c{k}_p{p}", str(k + 1)] + [fmt(x) for x in ab[:, j]], row
            row += 1
    assert lines[row].split("\t") == ["Unknown", "0"] + [fmt(x + 5) for x in noise]


def test_config0_plumbing_on_the_cpu(tmp_path):
    """BASELINE.json configs[0] / SURVEY.md §8d S1: 100,000 read pairs x 36,120 paths in 2,177 clusters, `-i transcripts`,
    the CPU restatement with 4 threads, no GPU — the driver's ordering, the file formats and the writers' invariants:
    dump -> reader (clusters by descending size) -> estimates -> rpvg.txt."""
    batch = synth.generate(seed=1, num_clusters=2177, total_paths=36120, total_reads=100000)
    probs, info = str(tmp_path / "s1_probs.txt.gz"), str(tmp_path / "s1_info.tsv")
    rio.write_batch_files(batch, probs, info)
    back = rio.read_batch_files(probs, info)
    nonempty = np.diff(batch.cluster_row_off.astype(np.int64)) > 0  # clusters without reads never reach the dump
    num_paths = int(np.diff(batch.cluster_path_off.astype(np.int64))[nonempty].sum())
    assert back.num_clusters == int(nonempty.sum()) and back.num_paths == num_paths and back.total_reads == 100000
    # ClusterID = rank by size: the dump does not carry the number of alignment-path lists the reference ranks by
    # (src/main.cpp:811-827), the reader ranks by read count
    reads = np.add.reduceat(back.row_count.astype(np.int64), back.cluster_row_off[:-1].astype(np.int64))
    assert np.all(reads[:-1] >= reads[1:])
    params = make_params()
    prefix = str(tmp_path / "s1")
    with pyoracle.RawRun("transcripts", params, back, 4) as run:
        rio.write_estimates(probs, info, "transcripts", params, run.view, prefix, unaligned_read_count=0)
        est = run.estimates
    # the invariant the reference's writers assert (src/threaded_output_writer.cpp:327-328)
    for e in est:
        if e.total_count > 0:
            assert abs(e.abundances.sum() + e.noise_count - e.total_count) <= 1e-9 * e.total_count
    assert sum(e.total_count for e in est) == 100000
    lines = open(prefix + ".txt").read().splitlines()
    assert lines[0] == "Name\tClusterID\tLength\tEffectiveLength\tReadCount\tTPM"
    assert len(lines) == 1 + num_paths + 1 and lines[-1].startswith("Unknown\t0\t0\t0\t")
    cols = [l.split("\t") for l in lines[1:-1]]
    assert abs(sum(float(c[5]) for c in cols) - 1e6) < 1.0  # TPM
    assert abs(sum(float(c[4]) for c in cols) + float(lines[-1].split("\t")[4]) - 100000) < 1e-2  # read counts
    assert [int(c[1]) for c in cols] == sorted(int(c[1]) for c in cols) and int(cols[-1][1]) == back.num_clusters


# ---- the one data file the reference holds for this path: example/pantranscriptome.txt.gz -----------------------------
# (tests/golden/make_info_fixture.py derives the fixture in the build container; the file itself does not travel)

def _info_fixture():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "info_example_pantranscriptome.json")) as f:
        return json.load(f)


def test_info_fixture_has_the_shape_the_survey_quotes():
    doc = _info_fixture()
    assert doc["num_paths"] == 36120 and doc["num_transcripts"] == 2177
    assert doc["hsts_per_transcript"] == dict(p50=8.0, p90=39.0, p99=124.0, max=648)
    assert round(doc["haplotypes_per_hst"]["mean"]) == 48 and doc["haplotypes_per_hst"]["max"] == 808
    assert len(doc["first_records"]) == 20 and len(doc["last_records"]) == 20


@pytest.mark.skipif(not os.path.exists("/root/reference/example/pantranscriptome.txt.gz"), reason="the reference checkout is not here")
def test_info_parser_reproduces_the_fixture_of_the_reference_example():
    from tests.golden import make_info_fixture as fx
    doc = _info_fixture()
    table = [tuple(r) for r in rio.info_table("/root/reference/example/pantranscriptome.txt.gz", parse_haplotype_ids=True)]
    assert fx.table_hash(table) == doc["sha256_of_parsed_table"]
    assert [list(r) for r in table[:20]] == doc["first_records"] and [list(r) for r in table[-20:]] == doc["last_records"]
    # without haplotype ids (`-i transcripts`, src/main.cpp:337): counted only
    counted = rio.info_table("/root/reference/example/pantranscriptome.txt.gz", parse_haplotype_ids=False)
    assert [r[3] for r in counted] == [r[3] for r in table] and all(not r[4] for r in counted)


def test_info_parser_on_records_rebuilt_from_the_fixture(tmp_path):
    """Travels: a `-f` file written from the fixture's first records parses back to them (dense ids renumbered in
    first-seen order, src/main.cpp:315-316,325-333)."""
    doc = _info_fixture()
    records = doc["first_records"]
    path = tmp_path / "info.txt"
    with open(path, "w") as f:
        f.write("Name\tLength\tTranscript\tHaplotypes\n")
        for key, name, group, count, ids in records:
            f.write(f"{key}\t1000\tT{group}\t{','.join('h%d' % i for i in ids)}\n")
    table = rio.info_table(str(path), parse_haplotype_ids=True)
    assert [r[0] for r in table] == [r[0] for r in records]
    groups, haps = {}, {}
    for key, name, group, count, ids in records:  # file order = name order here
        groups.setdefault(group, len(groups))
        for i in ids:
            haps.setdefault(i, len(haps))
    for got, (key, name, group, count, ids) in zip(table, records):
        assert got[2] == groups[group] and got[3] == len(ids) and got[4] == sorted(haps[i] for i in ids)


def test_s1_generator_has_the_shape_of_the_reference_example():
    """configs[0] / S1 restates the example with generated clusters: 36 120 paths in 2 177 groups whose sizes follow the
    example's HSTs-per-transcript distribution (a lognormal fit: the bulk within a third, the tail within 3x)."""
    doc = _info_fixture()
    batch = synth.generate(seed=1, num_clusters=doc["num_transcripts"], total_paths=doc["num_paths"], total_reads=100000)
    sizes = np.diff(batch.cluster_path_off.astype(np.int64))
    assert sizes.sum() == doc["num_paths"] and len(sizes) == doc["num_transcripts"]
    want = doc["hsts_per_transcript"]
    for q, key in ((50, "p50"), (90, "p90"), (99, "p99")):
        assert abs(np.percentile(sizes, q) - want[key]) <= 0.35 * want[key], (key, np.percentile(sizes, q), want[key])
    assert want["max"] / 3 <= sizes.max() <= want["max"] * 3


def test_dump_with_rank_keys_replays_in_the_reference_order(tmp_path):
    """The reference numbers and seeds clusters by their rank in descending (number of alignment-path lists, PathClusters
    index) order (src/main.cpp:811-827,849,976); its dump holds neither.  A producer that knows them writes "# <lists>
    <index>" as a block's marker line: the reader then ranks by that key instead of by read count."""
    import gzip
    batch = ClusterBatch.from_clusters(small_cases.make_batch_clusters(4411, n_clusters=12, with_empty=False))
    K = batch.num_clusters
    rng = np.random.default_rng(9)
    lists = rng.integers(1, 50, size=K).astype(np.uint64)
    lists[3] = lists[7]  # a tie: the larger cluster index goes first (reverse sort of (lists, index) pairs)
    index = rng.permutation(K).astype(np.uint64)
    probs, info = str(tmp_path / "ranked_probs.txt.gz"), str(tmp_path / "ranked_info.tsv")
    rio.write_batch_files(batch, probs, info, num_align_lists=lists, cluster_index=index)
    markers = [line for line in gzip.open(probs, "rt").read().splitlines() if line.startswith("#")]
    assert markers == [f"# {int(lists[k])} {int(index[k])}" for k in range(K)]
    back = rio.read_batch_files(probs, info)
    expected = sorted(range(K), key=lambda k: (int(lists[k]), int(index[k])), reverse=True)
    rows = np.diff(batch.cluster_row_off.astype(np.int64))
    reads = np.add.reduceat(batch.row_count.astype(np.int64), batch.cluster_row_off[:-1].astype(np.int64))
    back_rows = np.diff(back.cluster_row_off.astype(np.int64))
    back_reads = np.add.reduceat(back.row_count.astype(np.int64), back.cluster_row_off[:-1].astype(np.int64))
    assert back_rows.tolist() == [int(rows[k]) for k in expected] and back_reads.tolist() == [int(reads[k]) for k in expected]
    # a dump without keys (the reference's own) still ranks by read count
    plain = str(tmp_path / "plain_probs.txt.gz")
    rio.write_batch_files(batch, plain, info)
    by_reads = rio.read_batch_files(plain, info)
    r = np.add.reduceat(by_reads.row_count.astype(np.int64), by_reads.cluster_row_off[:-1].astype(np.int64))
    assert np.all(r[:-1] >= r[1:])
