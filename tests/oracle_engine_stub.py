"""Stand-in for rpvg_amd.engine in the CPU test of bench.py's launcher (tests/test_distributed_cpu.py): the same
Engine surface bench.py drives, with the CPU oracle doing the per-rank compute.  Test-only — bench.py loads it only
when RPVG_BENCH_ENGINE names it; the product engine has no CPU path."""
import time

from oracle import pyoracle

_STATS = ("em_sparse_ms", "em_sparse_launches", "em_sparse_alg_bytes", "em_dense_ms", "em_dense_launches", "em_dense_alg_bytes",
          "loglik_ms", "loglik_launches", "loglik_evals", "build_ms", "build_launches", "h2d_ms", "h2d_bytes", "em_iterations_total",
          "search_pairs_possible", "search_pairs_table", "search_pairs_kept")


class Prepared:
    def __init__(self, batch):
        self.batch = batch


class Engine:
    def __init__(self, device=0, uploader=False):
        self.device = device

    def close(self):
        pass

    def prepare(self, batch, per_cluster=False):
        return Prepared(batch)

    def run(self, model, params, prepared):
        t0 = time.perf_counter()
        est, _ = pyoracle.run(model, params, prepared.batch, 1)
        return est, time.perf_counter() - t0

    def run_raw(self, model, params, prepared):
        return self.run(model, params, prepared)[1]

    def stats(self):
        return {k: 0.0 for k in _STATS}

    def reset_stats(self):
        pass
