#!/usr/bin/env python3
"""Host timeline of ONE configs[2] data set through the batch pipeline, cut into parts (bench.py, measure_single_dataset):
RPVG_AMD_TIMELINE=1 python tools/single_dataset_timeline.py [fractions, e.g. 0.5,0.5] 2> timeline.txt
prints the wall time of every pass; the last pass's phases are the ones to read (the library prints every phase with its thread)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rpvg_amd import engine as eng_mod, hip, synth  # noqa: E402
from rpvg_amd.batch import make_params  # noqa: E402

fractions = [float(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0.5,0.5").split(",")]
workers = int(os.environ.get("WORKERS", "0"))
batch = synth.generate(seed=3, num_clusters=5000, total_paths=200000, total_reads=10000000)
arrays = bench.copied_arrays(batch)
for a in arrays:
    if a.nbytes >= bench.PAGE_LOCK_MIN_BYTES:
        hip.host_register(a)
pipe = eng_mod.Pipeline("haplotype-transcripts", make_params(), 0, workers=workers)
parts = bench.dataset_parts(batch, fractions)
for slot, part in enumerate(parts):
    pipe.prepare_slot(slot, part)
for rep in range(8):
    print(f"[timeline] pass {rep} begins", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for slot, part in enumerate(bench.dataset_parts(batch, fractions)):
        pipe.submit(part, slot, compact=True, narrow=True)
    pipe.wait()
    print(f"pass {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
pipe.close()
