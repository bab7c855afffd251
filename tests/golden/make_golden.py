"""Generates tests/golden/oracle_<model>.json: seeded small clusters (tests/small_cases.py) and the
estimates the CPU oracle produces for them, after checking the oracle against the independent numpy
restatement (oracle/np_oracle.py) on the same clusters.  The reference itself cannot be built in this
image, so these vectors pin the ORACLE ("parity unpinned" wrt the reference binary).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import np_oracle, pyoracle  # noqa: E402
from rpvg_amd.batch import ClusterBatch, make_params  # noqa: E402
from tests import small_cases  # noqa: E402

SEEDS = {"transcripts": 9001, "haplotype-transcripts": 9002, "haplotypes": 9003}


def main():
    for model, seed in SEEDS.items():
        clusters = small_cases.make_batch_clusters(seed, n_clusters=5, max_reads=250)
        params = dict()
        est, _ = pyoracle.run(model, make_params(**params), ClusterBatch.from_clusters(clusters), 1)
        out = []
        for cl, e in zip(clusters, est):
            keyed = e.keyed()
            if model == "transcripts":
                ref = np_oracle.estimate_transcripts(cl["paths"], cl["rows"])
                assert e.em_iters == ref["iters"]
                assert small_cases.rel_close(e.abundances, ref["abund"], rel=1e-9)
            elif model == "haplotype-transcripts":
                ref = np_oracle.estimate_haplotype_transcripts(cl["paths"], cl["rows"])
                assert set(keyed) == set(ref["keyed"])
                for k, (p, a) in ref["keyed"].items():
                    assert small_cases.rel_close(keyed[k][0], p, rel=1e-9) and small_cases.rel_close(keyed[k][1], a, rel=1e-8)
            else:
                ref = np_oracle.estimate_haplotypes(cl["paths"], cl["rows"], 2)
                assert set(keyed) == set(ref["keyed"])
            out.append(dict(sets=[[list(k), v[0], list(v[1])] for k, v in sorted(keyed.items())],
                            noise_count=e.noise_count, total_count=e.total_count, em_iters=e.em_iters))
        doc = dict(model=model, seed=seed, params=params,
                   clusters=[dict(paths=c["paths"], rows=[[r[0], r[1], [[g[0], g[1]] for g in r[2]]] for r in c["rows"]])
                             for c in clusters],
                   estimates=out)
        path = os.path.join(HERE, f"oracle_{model}.json")
        with open(path, "w") as f:
            json.dump(doc, f)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
