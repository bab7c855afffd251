#!/usr/bin/env python3
"""VALU instructions per pair-row evaluation and VALU busy fraction of the diploid search kernel from a PMC listing of
tools/pmc_kernels.py (profiles/r02/pmc_s3_search_kernels.txt) and the evaluations of one launch (bench.py's
roofline_search.evals_per_step of a one-lane run = one launch):

  python tools/pmc_search_summary.py <listing> <evals per launch> <out.json> [commit] [kernel] [launches]

launches: launches of the kernel in one PMC pass (the one-lane bench with --steps 2 --warmup 1 runs the search 11 times: start-up,
warm-up, timed, the upload leg and the decoded run); without it, from SQ_WAVES and round 2's 6 621 items of four waves.
"""
import json
import sys

listing, evals, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
commit = sys.argv[4] if len(sys.argv) > 4 else ""
kernel = sys.argv[5] if len(sys.argv) > 5 else "pairTile2Kernel"
given_launches = float(sys.argv[6]) if len(sys.argv) > 6 else None
counters, current = {}, None
for line in open(listing):
    if not line.startswith(" "):
        current = line.strip()
    elif current == kernel:
        name, value = line.split()
        counters[name] = float(value)
waves_per_launch = 4 * 6621  # 256-thread workgroups of the S3 batch (bench.py's configs[2] workload: 6 621 (matrix, chunk) items)
launches = given_launches if given_launches else counters.get("DISPATCHES_PER_PASS") or counters["SQ_WAVES"] / waves_per_launch
simds, xcds = 256 * 4, 8
# GRBM_GUI_ACTIVE is summed over the XCDs; a wave64 VALU instruction occupies its SIMD for 4 cycles
kernel_cycles = counters["GRBM_GUI_ACTIVE"] / xcds
summary = dict(kernel=kernel, launches=launches, evals_per_launch=evals,
               valu_instructions_per_eval=counters["SQ_INSTS_VALU"] * 64 / (launches * evals),
               valu_busy=counters["SQ_INSTS_VALU"] * 4 / (simds * kernel_cycles),
               lds_bank_conflict_share=counters.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, counters.get("SQ_LDS_IDX_ACTIVE", 0.0)),
               kernel_ms_at_2p4_ghz=kernel_cycles / launches / 2.4e6,
               source=listing, commit=commit, note="rocprofv3 --pmc passes, one host lane, every launch a whole batch (tools/refresh_profiles_r05.sh)")
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary))
