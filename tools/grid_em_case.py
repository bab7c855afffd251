#!/usr/bin/env python3
"""One large cluster through rpvg_hip_em_solve with a fixed iteration budget — the subject of a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --stats -- python tools/grid_em_case.py ROWS PATHS PER_ROW [ITERATIONS]`): per-kernel times of
the whole-GPU EM route (rpvg_amd/csrc/em_grid.hip)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rpvg_amd import hip
from tests import large_cases

rows, paths, per_row = (int(x) for x in sys.argv[1:4])
its = int(sys.argv[4]) if len(sys.argv) > 4 else 200
b = large_cases.cluster_batch(rows, paths, per_row, seed=7)
ctx = hip.Context(0)
dev = ctx.upload(b)
ctx.em_solve(dev, [0], [list(range(paths))], max_em_its=10, max_rel_em_conv=-1.0)
ctx.reset_stats()
_, _, _, done = ctx.em_solve(dev, [0], [list(range(paths))], max_em_its=its, max_rel_em_conv=-1.0)
st = ctx.stats()
for name, ks in st["em_kernel"].items():
    if ks["launches"] and ks["iterations"]:
        print(f"{name}: {int(done[0])} iterations, {ks['ms'] * 1e3 / int(done[0]):.3f} us per iteration")
if st["em_dense_launches"]:
    print(f"dense route: {int(done[0])} iterations, {st['em_dense_ms'] * 1e3 / int(done[0]):.3f} us per streaming pass")
dev.free()
ctx.close()
