#!/usr/bin/env python3
"""Experiment: throughput with two batches in flight on one GPU (two engines, each with its two host lanes, each on
its own copy of the bench batch, driven by two Python threads) against one batch at a time.  bench.py measures the
latter: a step there is one estimateBatch() call that returns before the next one starts."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rpvg_amd  # noqa: E402,F401
from rpvg_amd import engine as eng_mod, synth  # noqa: E402
from rpvg_amd.batch import make_params  # noqa: E402

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
model = "haplotype-transcripts"
params = make_params()
batch = synth.generate(seed=3)
engines = [eng_mod.Engine(0), eng_mod.Engine(0)]
prepared = [e.prepare(batch) for e in engines]
for e, p in zip(engines, prepared):
    for _ in range(4):
        e.run_raw(model, params, p)

t0 = time.perf_counter()
for _ in range(steps):
    engines[0].run_raw(model, params, prepared[0])
one = (time.perf_counter() - t0) / steps


def drive(k):
    for _ in range(steps // 2):
        engines[k].run_raw(model, params, prepared[k])


threads = [threading.Thread(target=drive, args=(k,)) for k in range(2)]
t0 = time.perf_counter()
for t in threads:
    t.start()
for t in threads:
    t.join()
two = (time.perf_counter() - t0) / (2 * (steps // 2))
print(f"one batch at a time: {one * 1e3:.2f} ms per batch; two in flight: {two * 1e3:.2f} ms per batch "
      f"({batch.total_reads / two / 1e6:.0f} M read pairs/s)")
