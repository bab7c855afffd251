#!/usr/bin/env python3
"""Re-runs one case of tests/fuzz_parity.py (seed as argument) and prints, for every cluster that differs, the group
sets with posteriors / abundances of both sides and the EM iteration counts."""
import sys

sys.path.insert(0, ".")
from rpvg_amd import engine as eng_mod  # noqa: E402
from rpvg_amd.batch import make_params  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tests import fuzz_parity  # noqa: E402

seed = int(sys.argv[1])
case = fuzz_parity.draw_case(seed, only_gibbs=("gibbs" in sys.argv))
print(case["model"], case["kw"], "clusters", case["batch"].num_clusters, "shape", case["shape"])
params = make_params(**case["kw"])
eng = eng_mod.Engine(0)
ref, _ = pyoracle.run(case["model"], params, case["batch"], 32)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1):
    got, _ = eng.run(case["model"], params, eng.prepare(case["batch"]))
    problems = fuzz_parity.compare(got, ref)
    print("run", rep, "problems", len(problems))
    for p in problems[:10]:
        print("   ", p)
    bad = sorted({int(p.split()[1].rstrip(":")) for p in problems if p.startswith("cluster")})
    for k in bad[:3]:
        g, r = got[k], ref[k]
        print("cluster", k, "rows", int(case["batch"].cluster_row_off[k + 1] - case["batch"].cluster_row_off[k]), "paths",
              int(case["batch"].cluster_path_off[k + 1] - case["batch"].cluster_path_off[k]), "total", g.total_count, r.total_count, "noise", g.noise_count, r.noise_count)
        print("  gpu em", list(zip(g.em_cols, g.em_iters))[:12])
        print("  ref em", list(zip(r.em_cols, r.em_iters))[:12])
        gk, rk = g.keyed(), r.keyed()
        if set(gk) != set(rk):
            print("  gpu sets", sorted(gk.items())[:12])
            print("  ref sets", sorted(rk.items())[:12])
        for key in list(rk)[:400]:
            if key in gk and (abs(gk[key][0] - rk[key][0]) > 1e-9 * max(1e-300, abs(rk[key][0])) or any(abs(a - b) > 1e-7 * max(1e-12, abs(b)) for a, b in zip(gk[key][1], rk[key][1]))):
                print("  set", key, "post", gk[key][0], rk[key][0], "abund", gk[key][1], rk[key][1])
