#!/bin/bash
# Re-measures the round-6 bench lines, kernel profiles, PMC passes and parity sweeps into gpurun_out/refresh/ (copy what is kept to
# profiles/r06/).  Run through gpurun from the repo root:  bash tools/refresh_profiles_r06.sh <commit>
# (<commit> = git rev-parse --short HEAD of the tree that is pushed: the GPU box has no .git; it is written into every JSON)
commit=${1:-unknown}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd $R
stamp() {  # adds the commit to a JSON line file
  python - "$1" "$commit" <<'PY'
import json, sys
path, commit = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    d["commit"] = commit
    open(path, "w").write(json.dumps(d) + "\n")
except Exception as e:
    print(path, "NOT STAMPED", e)
PY
}
# the driver's line (everything inside: one batch at a time, the pipeline, one data set, host bound, configs[4], the drop-in path, dense EM, CPU baseline)
timeout 900 python bench.py 2>$out/bench_s3.err | tail -1 > $out/bench_s3_n1.json; stamp $out/bench_s3_n1.json
timeout 900 python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_200_steps.json; stamp $out/bench_s3_n1_200_steps.json
ab() {  # name, env...: the pipeline only, 120 steps
  name=$1; shift
  env "$@" RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 900 python bench.py --steps 120 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_120_steps_$name.json; stamp $out/bench_s3_n1_120_steps_$name.json
}
ab default RPVG_X=1
ab wide_uploads RPVG_BENCH_WIDE_UPLOADS=1
ab default_again RPVG_X=2
for parts in 1 0.5,0.5 0.34,0.33,0.33; do
  RPVG_BENCH_PARTS=$parts RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_single_dataset_parts_$parts.json; stamp $out/bench_s3_n1_single_dataset_parts_$parts.json
done
timeout 900 python bench.py --workload c2 --steps 4 --warmup 1 2>$out/bench_c2.err | tail -1 > $out/bench_c2_n1.json; stamp $out/bench_c2_n1.json
timeout 900 python bench.py --workload s5 --steps 40 --warmup 6 2>$out/bench_s5.err | tail -1 > $out/bench_s5_n1.json; stamp $out/bench_s5_n1.json
RPVG_HIP_NO_FUSED_DENSE=1 timeout 900 python bench.py --workload c2 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_c2_n1_two_step_build.json; stamp $out/bench_c2_n1_two_step_build.json
RPVG_AMD_PORTABLE_GENERATORS=1 RPVG_BENCH_NO_HOST_BOUND=1 RPVG_BENCH_NO_SINGLE=1 timeout 900 python bench.py --workload s5 --steps 40 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s5_n1_portable_generators.json; stamp $out/bench_s5_n1_portable_generators.json
(echo "# tree $commit: configs[4], one resident batch, estimateBatch one call at a time on ONE host lane, OMP_WAIT_POLICY=passive (tools/s5_host_profile.py)"; OMP_WAIT_POLICY=passive RPVG_AMD_SINGLE_LANE=1 RPVG_AMD_TRACE=1 RPVG_AMD_TRACE_CPU=1 timeout 600 python tools/s5_host_profile.py 12) > $out/s5_host_cpu_by_phase.txt 2>/dev/null
(echo "# the same with libgomp's default wait policy"; RPVG_AMD_SINGLE_LANE=1 timeout 600 python tools/s5_host_profile.py 12 | head -1) >> $out/s5_host_cpu_by_phase.txt 2>/dev/null
# the drop-in path: teams of 64 and 256, the combiner's slots, every call alone
for t in 64 256; do timeout 900 python bench.py --workload a1 --team $t --steps 8 2>/dev/null | tail -1 > $out/bench_a1_n1_team_$t.json; stamp $out/bench_a1_n1_team_$t.json; done
RPVG_AMD_COMBINE_SLOTS=1 timeout 900 python bench.py --workload a1 --team 64 --steps 8 2>/dev/null | tail -1 > $out/bench_a1_n1_team_64_one_slot.json; stamp $out/bench_a1_n1_team_64_one_slot.json
RPVG_AMD_DEVICE_SOURCE_COLUMNS=1 timeout 900 python bench.py --workload a1 --team 64 --steps 8 2>/dev/null | tail -1 > $out/bench_a1_n1_team_64_device_columns.json; stamp $out/bench_a1_n1_team_64_device_columns.json
RPVG_AMD_NO_COMBINER=1 timeout 900 python bench.py --workload a1 --team 64 --steps 1 --warmup 1 2>/dev/null | tail -1 > $out/bench_a1_n1_team_64_no_combiner.json; stamp $out/bench_a1_n1_team_64_no_combiner.json
RPVG_AMD_TRACE=1 timeout 600 python bench.py --workload a1 --team 64 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/a1_phase_totals.txt
RPVG_AMD_TIMELINE=1 timeout 600 python bench.py --workload a1 --team 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 > /dev/null | grep "^\[timeline\]" | grep -v "flatten the cluster" > $out/a1_host_timeline.txt
(echo "# tree $commit, one MI355X box, team of 64: python bench.py --workload a1 --team 64 traced (RPVG_AMD_TRACE / RPVG_AMD_TIMELINE)"; python tools/a1_phase_breakdown.py $out/a1_phase_totals.txt $out/a1_host_timeline.txt) > $out/a1_phase_breakdown.txt
rm -f $out/a1_phase_totals.txt $out/a1_host_timeline.txt
python tools/em_iter_latency.py 20000 > $out/em_iteration_latency.txt 2>&1
cd /tmp; export TMPDIR=/tmp
prof() {  # name, bench args...
  name=$1; shift
  RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python $R/bench.py "$@" --no-cpu-baseline 2>$out/prof_$name.err | tail -1 > $out/bench_${name}_n1_profiled.json
  cp $out/prof_$name/*/*kernel_stats.csv $out/rocprofv3_${name}_kernel_stats.csv; rm -rf $out/prof_$name
  stamp $out/bench_${name}_n1_profiled.json
}
prof s3 --steps 40 --warmup 8
prof a1 --workload a1 --team 64 --steps 3 --warmup 1
prof c2 --workload c2 --steps 4 --warmup 1
RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/gpu_busy_union.py $out/prof_tl 16 > $out/gpu_busy_s3_pipeline.txt 2>&1; rm -rf $out/prof_tl
for lanes in two one; do
  extra=""; [ $lanes = one ] && extra="RPVG_AMD_SINGLE_LANE=1"
  env $extra RPVG_BENCH_NO_PIPELINE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl1 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/kernel_timeline.py $out/prof_tl1 0 > $out/kernel_timeline_s3_one_batch_${lanes}_lanes.txt 2>&1; rm -rf $out/prof_tl1
done
# PMC passes (each in its own run: counter slots; --kernel-trace only).  One host lane, the pipeline only: every launch holds a whole batch.
pmc() {  # dir, counters, bench args...
  d=$1; c=$2; shift 2
  RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$d -- python $R/bench.py "$@" --no-cpu-baseline > $out/$d.log 2>&1
}
pmc pmc_s3_fetch FETCH_SIZE --steps 2 --warmup 1
pmc pmc_s3_write WRITE_SIZE --steps 2 --warmup 1
cd $R
python tools/pmc_traffic.py --fetch-dir $out/pmc_s3_fetch --write-dir $out/pmc_s3_write --kernel emSparseKernel,emRegisterKernel --steps 1 --commit $commit \
  --command "RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_SINGLE=1 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (two separate passes; every launch holds the EM problems of a whole batch; per-launch figures only)" \
  --out $out/pmc_traffic_s3.json > /dev/null
rm -rf $out/pmc_s3_fetch $out/pmc_s3_write $out/*.log
# parity sweeps against the oracle on this tree
timeout 1500 python -m tests.fuzz_parity 200 61000 > $out/general_200_from_61000.txt 2>&1
timeout 900 python -m tests.fuzz_parity 60 62000 gibbs > $out/gibbs_60_from_62000.txt 2>&1
RPVG_FUZZ_TEAM=12 timeout 1500 python -m tests.fuzz_parity 150 63000 > $out/general_150_from_63000_through_estimate_team_of_12.txt 2>&1
echo $commit > $out/COMMIT
for f in bench_c2_n1_two_step_build bench_s5_n1_portable_generators bench_a1_n1_team_64_device_columns bench_s3_n1 bench_s3_n1_200_steps bench_s3_n1_120_steps_default bench_s3_n1_120_steps_wide_uploads bench_s3_n1_120_steps_default_again bench_c2_n1 bench_s5_n1 bench_a1_n1_team_64 bench_a1_n1_team_256 bench_a1_n1_team_64_one_slot bench_a1_n1_team_64_no_combiner bench_s3_n1_profiled bench_a1_n1_profiled; do python - <<PY
import json
try:
    d=json.loads(open("$out/$f.json").read())
    print("$f", round(d["ms_per_step"],2), round((d["value"] or 0)/1e6,1), d.get("single_dataset_ms"), d.get("host_cpu_ms_per_step"), d.get("cpu_baseline",{}).get("value"))
except Exception as e:
    print("$f FAILED", e)
PY
done
tail -2 $out/general_200_from_61000.txt $out/gibbs_60_from_62000.txt $out/general_150_from_63000_through_estimate_team_of_12.txt
ls $out
