#!/bin/bash
# Re-measures the round-2 bench lines, kernel profiles and PMC passes into gpurun_out/refresh/ (copy what is kept to
# profiles/r02/).  Run through gpurun from the repo root.
out=/root/repo/gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd /root/repo
timeout 600 python bench.py --steps 20 --warmup 5 2>$out/bench_s3.err | tail -1 > $out/bench_s3_n1.json
RPVG_HIP_PAIR_TILES=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_sequential_search.json
RPVG_HIP_NO_COLLAPSE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s3_n1_no_collapse.json
timeout 600 python bench.py --workload c2 --steps 4 --warmup 1 2>$out/bench_c2.err | tail -1 > $out/bench_c2_n1.json
timeout 600 python bench.py --workload s5 --steps 5 --warmup 1 2>$out/bench_s5.err | tail -1 > $out/bench_s5_n1.json
timeout 600 python bench.py --workload rows --steps 10 --warmup 2 2>$out/bench_rows.err | tail -1 > $out/bench_rows_n1.json
timeout 600 python bench.py --workload e2e --steps 10 --warmup 2 2>$out/bench_e2e.err | tail -1 > $out/bench_e2e_n1.json
cd /tmp; export TMPDIR=/tmp
prof() {  # name, bench args...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -- python /root/repo/bench.py "$@" --no-cpu-baseline 2>$out/prof_$name.err | tail -1 > $out/bench_${name}_n1_profiled.json
  cp $out/prof_$name/*/*kernel_stats.csv $out/rocprofv3_${name}_kernel_stats.csv; rm -rf $out/prof_$name
}
prof s3 --steps 20 --warmup 5
prof c2 --workload c2 --steps 4 --warmup 1
prof s5 --workload s5 --steps 5 --warmup 1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof_tl -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python /root/repo/tools/kernel_timeline.py $out/prof_tl 30 > $out/kernel_timeline_s3_one_step.txt; rm -rf $out/prof_tl
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
done
cd /root/repo
python tools/pmc_traffic.py --fetch-dir $out/pmc_s3_fetch --write-dir $out/pmc_s3_write --kernel emSparseKernel,emRegisterKernel --steps 3 \
  --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline (two separate passes; 3 passes of the hot path each: start-up, warmup, timed)" \
  --out $out/pmc_traffic_s3.json > /dev/null
python tools/pmc_traffic.py --fetch-dir $out/pmc_c2_fetch --write-dir $out/pmc_c2_write --kernel emDenseAccum --steps 100 --double-fetch --shape 1000000,2001,2002 \
  --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline (two separate passes; 2 steps x 50 EM iterations = 100 launches)" \
  --out $out/pmc_traffic_c2.json > /dev/null
python tools/pmc_kernels.py $out/pmc_search_1 $out/pmc_search_2 $out/pmc_search_3 $out/pmc_search_4 --kernel pairTile,resolveTable,Search,pairTable > $out/pmc_s3_search_kernels.txt
(cd $out && python /root/repo/tools/pmc_search_summary.py pmc_s3_search_kernels.txt 4573105636 pmc_search_s3.json > /dev/null)
python tools/pmc_kernels.py $out/pmc_s5_1 $out/pmc_s5_2 $out/pmc_s5_3 $out/pmc_s5_4 --kernel groupConditional,groupLoglik > $out/pmc_s5_conditional_kernels.txt
rm -rf $out/pmc_s3_fetch $out/pmc_s3_write $out/pmc_c2_fetch $out/pmc_c2_write $out/pmc_search_? $out/pmc_s5_?
for f in bench_s3_n1 bench_s3_n1_sequential_search bench_s3_n1_no_collapse bench_c2_n1 bench_s5_n1 bench_rows_n1 bench_e2e_n1 bench_s3_n1_profiled; do python - <<PY
import json
try:
    d=json.loads(open("$out/$f.json").read())
    print("$f", round(d["ms_per_step"],2), round(d["value"]/1e6,1), d.get("cpu_baseline",{}).get("value"))
except Exception as e:
    print("$f FAILED", e)
PY
done
ls $out
