import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from rpvg_amd import hip
from rpvg_amd.batch import ClusterBatch
from tests.test_hip_collapse import _near_identical_rows_cluster
for n in (3000, 30000, 100000):
    cl = _near_identical_rows_cluster(77, n)
    batch = ClusterBatch.from_clusters([cl])
    ctx = hip.Context(0)
    dev = ctx.upload(batch)
    ctx.em_solve(dev, [0], [[0, 1]], collapse_precision=1e-8)
    t = time.time(); r = ctx.em_solve(dev, [0], [[0, 1]], collapse_precision=1e-8); t1 = time.time() - t
    t = time.time(); r2 = ctx.em_solve(dev, [0], [[0, 1]]); t2 = time.time() - t
    print(n, "rows: em_solve with collapse %.1f ms, without %.1f ms, iterations %d / %d" % (t1 * 1e3, t2 * 1e3, r[3][0], r2[3][0]))
    dev.free(); ctx.close()
