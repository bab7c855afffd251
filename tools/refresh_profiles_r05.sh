#!/bin/bash
# Re-measures the round-5 bench lines, kernel profiles and PMC passes into gpurun_out/refresh/ (copy what is kept to
# profiles/r05/).  Run through gpurun from the repo root:  bash tools/refresh_profiles_r05.sh <commit>
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
# the driver's line (everything inside: one batch at a time, the pipeline, host bound, configs[4], dense EM, CPU baseline)
timeout 900 python bench.py 2>$out/bench_s3.err | tail -1 > $out/bench_s3_n1.json; stamp $out/bench_s3_n1.json
timeout 900 python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_200_steps.json; stamp $out/bench_s3_n1_200_steps.json
RPVG_AMD_PIPELINE_WORKERS=1 timeout 900 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_one_worker.json; stamp $out/bench_s3_n1_one_worker.json
RPVG_AMD_PIPELINE_WORKERS=2 timeout 900 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_two_workers.json; stamp $out/bench_s3_n1_two_workers.json
RPVG_AMD_HOST_SOURCE_GROUPS=1 timeout 900 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_host_source_groups.json; stamp $out/bench_s3_n1_host_source_groups.json
RPVG_HIP_SPIN_WAITS=1 timeout 900 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_spin_waits.json; stamp $out/bench_s3_n1_spin_waits.json
# the switches between two correct implementations of the second half of the round (docs/design/history-r05.md), one at a time, no one-batch leg
ab() {  # name, env...
  name=$1; shift
  env "$@" RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 900 python bench.py --steps 120 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_120_steps_$name.json; stamp $out/bench_s3_n1_120_steps_$name.json
}
ab default RPVG_X=1
ab pooled_main_queue RPVG_HIP_POOLED_MAIN_QUEUE=1
ab register_bins_in_five_launches RPVG_HIP_EM_REGISTER_LAUNCHES=5
ab collapse_before_search RPVG_HIP_COLLAPSE_BEFORE_SEARCH=1
ab mask_build RPVG_HIP_BUILD_MASKS=1
ab default_again RPVG_X=2
timeout 900 python bench.py --workload c2 --steps 4 --warmup 1 2>$out/bench_c2.err | tail -1 > $out/bench_c2_n1.json; stamp $out/bench_c2_n1.json
timeout 900 python bench.py --workload s5 --steps 40 --warmup 6 2>$out/bench_s5.err | tail -1 > $out/bench_s5_n1.json; stamp $out/bench_s5_n1.json
timeout 900 python bench.py --workload rows --steps 10 --warmup 2 2>$out/bench_rows.err | tail -1 > $out/bench_rows_n1.json; stamp $out/bench_rows_n1.json
timeout 900 python bench.py --workload e2e --steps 10 --warmup 2 2>$out/bench_e2e.err | tail -1 > $out/bench_e2e_n1.json; stamp $out/bench_e2e_n1.json
for t in 64 256; do timeout 900 python bench.py --workload a1 --team $t --steps 5 2>/dev/null | tail -1 > $out/bench_a1_n1_team_$t.json; stamp $out/bench_a1_n1_team_$t.json; done
RPVG_AMD_NO_COMBINER=1 timeout 900 python bench.py --workload a1 --team 64 --steps 1 --warmup 1 2>/dev/null | tail -1 > $out/bench_a1_n1_team_64_no_combiner.json; stamp $out/bench_a1_n1_team_64_no_combiner.json
python tools/em_iter_latency.py 20000 > $out/em_iteration_latency.txt 2>&1
cd /tmp; export TMPDIR=/tmp
prof() {  # name, env + bench args...
  name=$1; shift
  RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python $R/bench.py "$@" --no-cpu-baseline 2>$out/prof_$name.err | tail -1 > $out/bench_${name}_n1_profiled.json
  cp $out/prof_$name/*/*kernel_stats.csv $out/rocprofv3_${name}_kernel_stats.csv; rm -rf $out/prof_$name
  stamp $out/bench_${name}_n1_profiled.json
}
prof s3 --steps 40 --warmup 8
prof c2 --workload c2 --steps 4 --warmup 1
prof s5 --workload s5 --steps 10 --warmup 2
RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/gpu_busy_union.py $out/prof_tl 16 > $out/gpu_busy_s3_pipeline.txt 2>&1
python $R/tools/r05_pipe_timeline.py $out/prof_tl 14 25 > $out/kernel_timeline_s3_pipeline.txt 2>&1; rm -rf $out/prof_tl
RPVG_BENCH_NO_PIPELINE=1 RPVG_BENCH_NO_GIBBS_LINE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl1 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/kernel_timeline.py $out/prof_tl1 15 > $out/kernel_timeline_s3_one_batch.txt 2>&1; rm -rf $out/prof_tl1
# PMC passes (each in its own run: counter slots; --kernel-trace only).  One host lane, the pipeline only: every launch holds a whole batch.
pmc() {  # dir, counters, bench args...
  d=$1; c=$2; shift 2
  RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$d -- python $R/bench.py "$@" --no-cpu-baseline > $out/$d.log 2>&1
}
pmc pmc_s3_fetch FETCH_SIZE --steps 2 --warmup 1
pmc pmc_s3_write WRITE_SIZE --steps 2 --warmup 1
pmc pmc_c2_fetch FETCH_SIZE --workload c2 --steps 1 --warmup 1
pmc pmc_c2_write WRITE_SIZE --workload c2 --steps 1 --warmup 1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  pmc pmc_search_$i "$set" --steps 2 --warmup 1
done
cd $R
python tools/pmc_traffic.py --fetch-dir $out/pmc_s3_fetch --write-dir $out/pmc_s3_write --kernel emSparseKernel,emRegisterKernel --steps 1 --commit $commit \
  --command "RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_SINGLE=1 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (two separate passes; every launch holds the EM problems of a whole batch; per-launch figures only)" \
  --out $out/pmc_traffic_s3.json > /dev/null
python tools/pmc_traffic.py --fetch-dir $out/pmc_c2_fetch --write-dir $out/pmc_c2_write --kernel emDenseAccum --steps 150 --double-fetch --shape 1000000,2001,2002 --commit $commit \
  --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline (two separate passes; 3 estimator calls x 50 EM iterations = 150 launches)" \
  --out $out/pmc_traffic_c2.json > /dev/null
python tools/pmc_kernels.py $out/pmc_search_1 $out/pmc_search_2 $out/pmc_search_3 $out/pmc_search_4 --kernel pairTile,resolveTable,groupsBuild,fillSegments,subsetSelect,subsetMerge,sourceColumns,emRegister,emSparse,collapse,partitionRows,expandGroups,validateRows > $out/pmc_s3_throughput_kernels.txt
(cd $out && python $R/tools/pmc_search_summary.py pmc_s3_throughput_kernels.txt 4573105636 pmc_search_s3.json $commit pairTile2Kernel > /dev/null)
rm -rf $out/pmc_s3_fetch $out/pmc_s3_write $out/pmc_c2_fetch $out/pmc_c2_write $out/pmc_search_?
echo $commit > $out/COMMIT
for f in bench_s3_n1_120_steps_default bench_s3_n1_120_steps_pooled_main_queue bench_s3_n1_120_steps_register_bins_in_five_launches bench_s3_n1_120_steps_collapse_before_search bench_s3_n1_120_steps_mask_build bench_s3_n1_120_steps_default_again bench_s3_n1 bench_s3_n1_200_steps bench_s3_n1_one_worker bench_s3_n1_two_workers bench_s3_n1_host_source_groups bench_s3_n1_spin_waits bench_c2_n1 bench_s5_n1 bench_rows_n1 bench_e2e_n1 bench_a1_n1_team_64 bench_a1_n1_team_256 bench_a1_n1_team_64_no_combiner bench_s3_n1_profiled; do python - <<PY
import json
try:
    d=json.loads(open("$out/$f.json").read())
    print("$f", round(d["ms_per_step"],2), round((d["value"] or 0)/1e6,1), d.get("ms_per_step_resident"), d.get("host_cpu_ms_per_step"), d.get("cpu_baseline",{}).get("value"))
except Exception as e:
    print("$f FAILED", e)
PY
done
ls $out
