"""Generates tests/golden/info_example_pantranscriptome.json from the one data file the reference holds for this
path: example/pantranscriptome.txt.gz, the `-f` path info of its bundled example (36 120 haplotype-specific
transcripts of 2 177 transcripts; parsed by src/main.cpp:239-353).

The file is read twice — by the repo's `-f` parser (rpvg_amd/host/io/cluster_io.cpp, through the harness) and by the
independent restatement of src/main.cpp:239-353 below — the two must agree, and what travels is DERIVED data only:
counts, the quantiles SURVEY.md §8d quotes for the S1 workload, the first and last 20 parsed records in name order and
a hash of the whole parsed table.  Runs in the build container only (the reference does not travel).

    python tests/golden/make_info_fixture.py [/root/reference/example/pantranscriptome.txt.gz]
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

DEFAULT = "/root/reference/example/pantranscriptome.txt.gz"


def restated_parse(path):
    """src/main.cpp:239-353 for `-i haplotype-transcripts`: header starts with Name (old format: a Reference column before
    Haplotypes), transcript -> dense group id in first-seen order (:315-316), haplotype names -> dense source ids in
    first-seen order (:325-333), source_count = number of haplotypes listed."""
    group_of, hap_of, rows = {}, {}, {}
    old_format = False
    with gzip.open(path, "rt") as f:
        for lineno, line in enumerate(f):
            line = line.rstrip("\n")
            if not line:
                continue
            fields = line.split("\t")
            if lineno == 0:
                assert fields[0] == "Name"
                old_format = "Reference" in line
                continue
            name, transcript, haplotypes = fields[0], fields[2], fields[4 if old_format else 3]
            assert name not in rows
            group = group_of.setdefault(transcript, len(group_of))
            ids = sorted({hap_of.setdefault(h, len(hap_of)) for h in haplotypes.split(",") if h})
            rows[name] = (name, name, group, len([h for h in haplotypes.split(",") if h]), ids)
    return [rows[k] for k in sorted(rows)]


def table_hash(table):
    h = hashlib.sha256()
    for key, name, group, count, ids in table:
        h.update(f"{key}\t{name}\t{group}\t{count}\t{','.join(map(str, ids))}\n".encode())
    return h.hexdigest()


def summary(table):
    per_group = np.bincount(np.array([r[2] for r in table]))
    haps = np.array([r[3] for r in table])
    return dict(num_paths=len(table), num_transcripts=int(len(per_group)),
                hsts_per_transcript=dict(p50=float(np.percentile(per_group, 50)), p90=float(np.percentile(per_group, 90)),
                                         p99=float(np.percentile(per_group, 99)), max=int(per_group.max())),
                haplotypes_per_hst=dict(mean=float(haps.mean()), max=int(haps.max())),
                num_haplotype_ids=int(max(max(r[4]) for r in table if r[4]) + 1))


def main():
    from rpvg_amd import io
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    parsed = [tuple(r) for r in io.info_table(path, parse_haplotype_ids=True)]
    restated = restated_parse(path)
    assert len(parsed) == len(restated)
    assert all(p == tuple(r) for p, r in zip(parsed, restated)), "the repo's -f parser and the restatement of src/main.cpp:239-353 differ"
    doc = dict(source="example/pantranscriptome.txt.gz of the reference checkout (derived data only)", sha256_of_parsed_table=table_hash(parsed),
               first_records=[list(r) for r in parsed[:20]], last_records=[list(r) for r in parsed[-20:]], **summary(parsed))
    out = os.path.join(HERE, "info_example_pantranscriptome.json")
    with open(out, "w") as f:
        json.dump(doc, f)
    print(out, os.path.getsize(out), "bytes;", {k: doc[k] for k in ("num_paths", "num_transcripts", "hsts_per_transcript", "haplotypes_per_hst")})


if __name__ == "__main__":
    main()
