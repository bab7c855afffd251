#!/usr/bin/env python3
"""Latency of ONE EM iteration of a lone problem, per kernel variant: a single problem of the given shape runs a fixed
number of iterations (max_rel_em_conv = -1: never converges), alone on the GPU; the kernel's HIP-event span divided by the
iterations is the figure the EM tail of a batch is made of.

    python tools/em_iter_latency.py [iterations]
"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rpvg_amd import hip
from rpvg_amd.batch import ClusterBatch
from tests import large_cases

its = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(7)


def cluster(rows, paths, per_row):
    rs = []
    for r in range(rows):
        k = min(paths, per_row if r % 9 else per_row + 1)
        idx = sorted(rng.choice(paths, size=k, replace=False).tolist())
        noise = float(rng.choice([1e-4, 1e-3, 1e-2]))
        w = rng.random(k) + 0.1
        w = w / w.sum() * (1 - noise)
        order = np.argsort(w)
        rs.append((int(rng.integers(1, 5)), noise, [(float(w[i]) + 1e-7 * j, [idx[i]]) for j, i in enumerate(order)]))
    return dict(paths=[dict(group_id=0, source_count=1, source_ids=[0], effective_length=100.0) for _ in range(paths)], rows=rs)


shapes = [(17, 14, 1), (60, 15, 1), (120, 14, 1), (250, 15, 1), (17, 14, 6), (100, 24, 1), (60, 30, 2), (300, 40, 2), (2000, 40, 2),
          (20000, 200, 3), (200000, 400, 3), (1000000, 1000, 4), (8000, 400, 40), (200000, 2000, 100), (4000, 200, 200), (50000, 1000, 1000)]
ctx = hip.Context(0)
for rows, paths, per_row in shapes:
    # (the large ones vectorised: tests/large_cases.py)
    b = large_cases.cluster_batch(rows, paths, per_row, seed=7) if rows >= 4000 else ClusterBatch.from_clusters([cluster(rows, paths, per_row)])
    dev = ctx.upload(b)
    n = max(200 if rows >= 20000 else 50, its // max(1, rows // 50))
    ctx.em_solve(dev, [0], [list(range(paths))], max_em_its=10, max_rel_em_conv=-1.0)
    ctx.reset_stats()
    _, _, _, done = ctx.em_solve(dev, [0], [list(range(paths))], max_em_its=n, max_rel_em_conv=-1.0)
    st = ctx.stats()
    for name, ks in st["em_kernel"].items():
        if ks["launches"] and ks["iterations"]:
            gbs = ks["alg_bytes"] / 1e9 / (ks["ms"] / 1e3)
            print(f"rows {rows:7d} paths {paths:4d} entries/row {per_row:4d}: {name:28s} {int(done[0]):6d} iterations, {ks['ms'] * 1e3 / int(done[0]):8.3f} us per iteration, {gbs:8.1f} GB/s")
    if st["em_dense_launches"]:  # the dense route of the grid bin (em_dense.hip's kernels)
        gbs = st["em_dense_alg_bytes"] / 1e9 / (st["em_dense_ms"] / 1e3)
        print(f"rows {rows:7d} paths {paths:4d} entries/row {per_row:4d}: {'emDenseAccum[Wide]Kernel':28s} {int(done[0]):6d} iterations, {st['em_dense_ms'] * 1e3 / int(done[0]):8.3f} us per iteration (streaming pass only), {gbs:8.1f} GB/s")
    dev.free()
ctx.close()
