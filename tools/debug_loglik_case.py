#!/usr/bin/env python3
"""One cluster of a tests/fuzz_parity.py shape-0 case: single-column log-likelihoods on the device vs numpy."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import np_oracle  # noqa: E402
from rpvg_amd import hip  # noqa: E402
from rpvg_amd.batch import ClusterBatch  # noqa: E402
from tests import small_cases  # noqa: E402

seed, k = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
assert rng.integers(0, 3) == 0
clusters = small_cases.make_batch_clusters(seed, n_clusters=int(rng.integers(1, 200)), max_reads=int(rng.integers(25, 400)))
cl = clusters[k]
print("rows", len(cl["rows"]), "paths", len(cl["paths"]))
for row in cl["rows"][:40]:
    print("   count", row[0], "noise", row[1], "groups", [(round(p, 6), idx) for p, idx in row[2]][:4])
ctx = hip.Context(0)
dev = ctx.upload(ClusterBatch.from_clusters([cl]))
g, mult = np_oracle.source_groups(cl["paths"])
dg = ctx.groups(dev, [0], [g], True)
M, noise, counts = np_oracle.grouped_matrix(cl["rows"], g)
Mn = np_oracle.add_noise_and_normalize(M, noise)[:, :-1]
G = len(g)
want = np.array([np_oracle.set_loglik(Mn, noise, counts, (a,), 1) for a in range(G)])
got = dg.loglik([0] * G, [[a] for a in range(G)], 1.0)
print("want", want)
print("got ", got)
print("diff", got - want)
print("noise", noise, "counts", counts)
from oracle import pyoracle  # noqa: E402
mult = np.array(mult, dtype=float)
lf = np.log(mult / mult.sum())
for name, ll in (("numpy", want), ("gpu", got)):
    z = ll + lf
    p = np.exp(z - z.max())
    p /= p.sum()
    print(name, "posteriors", p)
sets, post = pyoracle.group_posteriors(Mn, noise, counts, [int(m) for m in mult], 1, bounded=False)
print("c++ oracle", sets, post)
print("groups", g, "mult", mult)
Mc, cc = np_oracle.read_collapse(np_oracle.add_noise_and_normalize(M, noise), counts, 1e-8)
print("rows before/after collapse", len(counts), len(cc))
wantc = np.array([float(cc @ np.log(Mc[:, -1] + Mc[:, a])) for a in range(G)])
z = wantc + lf
p = np.exp(z - z.max()); p /= p.sum()
print("numpy collapsed posteriors", p)
full = np_oracle.add_noise_and_normalize(M, noise)
order = np.lexsort(full.T[::-1])
for i in order[:]:
    print(counts[i], np.round(full[i], 10))
