#!/bin/bash
# Re-measures the round-4 bench lines, kernel profiles and PMC passes into gpurun_out/refresh/ (copy what is kept to
# profiles/r04/).  Run through gpurun from the repo root:  bash tools/refresh_profiles_r04.sh <commit>
# (<commit> = git rev-parse --short HEAD of the tree that is pushed: the GPU box has no .git; it is written into every JSON)
commit=${1:-unknown}
out=/root/repo/gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd /root/repo
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
timeout 600 python bench.py --steps 20 --warmup 5 2>$out/bench_s3.err | tail -1 > $out/bench_s3_n1.json; stamp $out/bench_s3_n1.json
RPVG_AMD_HOST_THREADS=4 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_host_threads_4.json; stamp $out/bench_s3_n1_host_threads_4.json
RPVG_HIP_NO_DEVICE_SUBSETS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_separate_calls.json; stamp $out/bench_s3_n1_separate_calls.json
RPVG_HIP_NO_DEVICE_SUBSETS=1 RPVG_AMD_HOST_THREADS=4 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_separate_calls_host_threads_4.json; stamp $out/bench_s3_n1_separate_calls_host_threads_4.json
RPVG_HIP_NO_COLLAPSE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_no_collapse.json; stamp $out/bench_s3_n1_no_collapse.json
timeout 600 python bench.py --workload c2 --steps 4 --warmup 1 2>$out/bench_c2.err | tail -1 > $out/bench_c2_n1.json; stamp $out/bench_c2_n1.json
timeout 600 python bench.py --workload s5 --steps 5 --warmup 1 2>$out/bench_s5.err | tail -1 > $out/bench_s5_n1.json; stamp $out/bench_s5_n1.json
timeout 600 python bench.py --workload rows --steps 10 --warmup 2 2>$out/bench_rows.err | tail -1 > $out/bench_rows_n1.json; stamp $out/bench_rows_n1.json
timeout 600 python bench.py --workload e2e --steps 10 --warmup 2 2>$out/bench_e2e.err | tail -1 > $out/bench_e2e_n1.json; stamp $out/bench_e2e_n1.json
python tools/em_iter_latency.py 20000 > $out/em_iteration_latency.txt 2>&1
timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_60_steps.json; stamp $out/bench_s3_n1_60_steps.json
RPVG_HIP_EM_LAUNCH_EARLY=1 RPVG_HIP_SEARCH_LAUNCH_EARLY=1 RPVG_HIP_EM_JOIN_ON_STREAM=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_60_steps_parked_streams.json; stamp $out/bench_s3_n1_60_steps_parked_streams.json
RPVG_HIP_COLLAPSE_LIBRARY_SORT=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_60_steps_library_sort.json; stamp $out/bench_s3_n1_60_steps_library_sort.json
RPVG_HIP_PAIR_TILES=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_60_steps_round2_tile_kernel.json; stamp $out/bench_s3_n1_60_steps_round2_tile_kernel.json
RPVG_HIP_BUILD_MASKS=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_60_steps_mask_build.json; stamp $out/bench_s3_n1_60_steps_mask_build.json
DBG="0 1 2 4 8 15 16 31" bash tools/r04_tile_debug.sh > $out/pair_tile_breakdown.txt 2>&1
cd /tmp; export TMPDIR=/tmp
prof() {  # name, bench args...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python /root/repo/bench.py "$@" --no-cpu-baseline 2>$out/prof_$name.err | tail -1 > $out/bench_${name}_n1_profiled.json
  cp $out/prof_$name/*/*kernel_stats.csv $out/rocprofv3_${name}_kernel_stats.csv; rm -rf $out/prof_$name
  stamp $out/bench_${name}_n1_profiled.json
}
prof s3 --steps 20 --warmup 5
prof c2 --workload c2 --steps 4 --warmup 1
prof s5 --workload s5 --steps 5 --warmup 1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_grid -- python /root/repo/tools/grid_em_case.py 200000 400 3 400 > $out/grid_em_200k.txt 2>&1
cp $out/prof_grid/*/*kernel_stats.csv $out/rocprofv3_grid_em_200k_kernel_stats.csv; rm -rf $out/prof_grid
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python /root/repo/tools/kernel_timeline.py $out/prof_tl 3 > $out/kernel_timeline_s3_one_step.txt
python /root/repo/tools/gpu_gaps.py $out/prof_tl > $out/gpu_gaps_s3.txt 2>&1; rm -rf $out/prof_tl
# PMC passes (each in its own run: counter slots; --kernel-trace only)
pmc() {  # dir, counters, bench args...
  d=$1; c=$2; shift 2
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$d -- python /root/repo/bench.py "$@" --no-cpu-baseline > $out/$d.log 2>&1
}
pmc pmc_s3_fetch FETCH_SIZE --steps 1 --warmup 1
pmc pmc_s3_write WRITE_SIZE --steps 1 --warmup 1
pmc pmc_c2_fetch FETCH_SIZE --workload c2 --steps 1 --warmup 1
pmc pmc_c2_write WRITE_SIZE --workload c2 --steps 1 --warmup 1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  RPVG_AMD_SINGLE_LANE=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_search_$i -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_search_$i.log 2>&1
  RPVG_AMD_SINGLE_LANE=1 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_s5_$i -- python /root/repo/bench.py --workload s5 --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_s5_$i.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_emlat_$i -- python /root/repo/tools/em_iter_latency.py 20000 > $out/pmc_emlat_$i.log 2>&1
done
cd /root/repo
# (the bench command with --steps 1 --warmup 1 runs the hot path 8 times: start-up, warmup, timed; upload leg: 2 warmups, timed, serial; the decoded run)
python tools/pmc_traffic.py --fetch-dir $out/pmc_s3_fetch --write-dir $out/pmc_s3_write --kernel emSparseKernel,emRegisterKernel --steps 8 --commit $commit \
  --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline (two separate passes)" \
  --out $out/pmc_traffic_s3.json > /dev/null
python tools/pmc_traffic.py --fetch-dir $out/pmc_c2_fetch --write-dir $out/pmc_c2_write --kernel emDenseAccum --steps 150 --double-fetch --shape 1000000,2001,2002 --commit $commit \
  --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline (two separate passes; 3 estimator calls x 50 EM iterations = 150 launches)" \
  --out $out/pmc_traffic_c2.json > /dev/null
python tools/pmc_kernels.py $out/pmc_search_1 $out/pmc_search_2 $out/pmc_search_3 $out/pmc_search_4 --kernel pairTile,resolveTable,Search,pairTable > $out/pmc_s3_search_kernels.txt
(cd $out && python /root/repo/tools/pmc_search_summary.py pmc_s3_search_kernels.txt 4573105636 pmc_search_s3.json $commit pairTile2Kernel 11 > /dev/null)
python tools/pmc_kernels.py $out/pmc_s5_1 $out/pmc_s5_2 $out/pmc_s5_3 $out/pmc_s5_4 --kernel groupConditional,groupLoglik,gibbs > $out/pmc_s5_conditional_kernels.txt
python tools/pmc_kernels.py $out/pmc_emlat_1 $out/pmc_emlat_2 $out/pmc_emlat_3 $out/pmc_emlat_4 --kernel emRegisterKernel,emSparseKernel > $out/pmc_em_iteration_latency.txt
rm -rf $out/pmc_s3_fetch $out/pmc_s3_write $out/pmc_c2_fetch $out/pmc_c2_write $out/pmc_search_? $out/pmc_s5_? $out/pmc_emlat_?
echo $commit > $out/COMMIT
for f in bench_s3_n1 bench_s3_n1_60_steps bench_s3_n1_60_steps_round2_tile_kernel bench_s3_n1_60_steps_mask_build bench_s3_n1_60_steps_parked_streams bench_s3_n1_60_steps_library_sort bench_s3_n1_host_threads_4 bench_s3_n1_separate_calls bench_s3_n1_separate_calls_host_threads_4 bench_s3_n1_no_collapse bench_c2_n1 bench_s5_n1 bench_rows_n1 bench_e2e_n1 bench_s3_n1_profiled; do python - <<PY
import json
try:
    d=json.loads(open("$out/$f.json").read())
    print("$f", round(d["ms_per_step"],2), round((d["value"] or 0)/1e6,1), d.get("ms_per_step_resident"), d.get("cpu_baseline",{}).get("value"))
except Exception as e:
    print("$f FAILED", e)
PY
done
ls $out
