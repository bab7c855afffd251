"""GPU end-to-end over the on-disk formats: `--write-probs` dump + `-f` info -> GPU estimators -> the
reference's result TSVs, against the same files written from the CPU oracle's estimates."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from rpvg_amd import io as rio, synth
from rpvg_amd.batch import make_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(path, n_key):
    rows = {}
    lines = open(path).read().splitlines()
    for line in lines[1:]:
        f = line.split("\t")
        rows[tuple(f[:n_key])] = [float(x) for x in f[n_key:]]
    return lines[0], rows


def _same(a, b, rel=1e-5, floor=1e-6):
    (ha, ra), (hb, rb) = a, b
    assert ha == hb
    assert set(ra) == set(rb)
    for k in ra:
        x, y = np.array(ra[k]), np.array(rb[k])
        assert np.all(np.abs(x - y) <= np.maximum(rel * np.maximum(np.abs(x), np.abs(y)), floor)), (k, x, y)


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("replay")
    batch = synth.generate(seed=41, num_clusters=40, total_paths=900, total_reads=40000)
    probs, info = str(d / "run_probs.txt.gz"), str(d / "info.tsv.gz")
    rio.write_batch_files(batch, probs, info)
    return dict(dir=d, probs=probs, info=info)


@pytest.mark.parametrize("model,keys", [("transcripts", [("", 2)]), ("strains", [("", 2)]),
                                        ("haplotype-transcripts", [("", 2), ("_joint", 3)]), ("haplotypes", [("", 3)])])
def test_replay_writes_the_reference_files(files, model, keys):
    params = make_params()
    batch = rio.read_batch_files(files["probs"], files["info"], parse_haplotype_ids=(model == "haplotype-transcripts"))
    gpu_prefix, cpu_prefix = str(files["dir"] / f"gpu_{model}"), str(files["dir"] / f"cpu_{model}")
    n = rio.replay(files["probs"], files["info"], model, params, gpu_prefix, unaligned_read_count=9)
    assert n == batch.num_clusters
    with pyoracle.RawRun(model, params, batch, 2) as run:
        rio.write_estimates(files["probs"], files["info"], model, params, run.view, cpu_prefix, unaligned_read_count=9)
    for suffix, n_key in keys:
        _same(_table(gpu_prefix + suffix + ".txt", n_key), _table(cpu_prefix + suffix + ".txt", n_key))


def test_replay_command_line(files):
    exe = os.path.join(ROOT, "rpvg_amd", "host", "rpvg_amd_replay")
    prefix = str(files["dir"] / "cli")
    out = subprocess.run([exe, "-p", files["probs"], "-f", files["info"], "-i", "haplotype-transcripts", "-o", prefix, "-n", "4",
                          "--gibbs-thin-its", "2", "-r", "5"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Inferred haplotype-transcripts estimates for 40 clusters" in out.stderr
    for suffix in (".txt", "_joint.txt", "_gibbs.txt.gz"):
        assert os.path.getsize(prefix + suffix) > 0
    bad = subprocess.run([exe, "-p", files["probs"], "-i", "haplotype-transcripts", "-o", prefix], capture_output=True, text=True, timeout=300)
    assert bad.returncode == 1 and "path info" in bad.stderr
