#!/bin/bash
# Re-measures what the device Gibbs sampler changed (configs[4], --workload s5) plus the default line at the same commit, into
# gpurun_out/refresh_s5/ (copy what is kept to profiles/r03/).  Run through gpurun from the repo root:
#   bash tools/refresh_profiles_r03_s5.sh <commit>
commit=${1:-unknown}
out=/root/repo/gpurun_out/refresh_s5
rm -rf $out; mkdir -p $out
cd /root/repo
stamp() {
  timeout 30 python - "$1" "$commit" <<'PY'
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
timeout 600 python bench.py --workload s5 --steps 10 --warmup 2 2>$out/bench_s5.err | tail -1 > $out/bench_s5_n1.json; stamp $out/bench_s5_n1.json
RPVG_AMD_HOST_GIBBS=1 timeout 600 python bench.py --workload s5 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s5_n1_host_sampler.json; stamp $out/bench_s5_n1_host_sampler.json
RPVG_AMD_HOST_THREADS=4 timeout 600 python bench.py --workload s5 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_s5_n1_host_threads_4.json; stamp $out/bench_s5_n1_host_threads_4.json
RPVG_HIP_GIBBS_DEBUG=1 RPVG_AMD_SINGLE_LANE=1 timeout 300 python bench.py --workload s5 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "rpvg_hip gibbs" | head -40 > $out/gibbs_rounds_s5_single_lane.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_s5 -- python /root/repo/bench.py --workload s5 --steps 5 --warmup 1 --no-cpu-baseline 2>$out/prof_s5.err | tail -1 > $out/bench_s5_n1_profiled.json
cp $out/prof_s5/*/*kernel_stats.csv $out/rocprofv3_s5_kernel_stats.csv
timeout 60 python /root/repo/tools/kernel_timeline.py $out/prof_s5 3 > $out/kernel_timeline_s5_one_step.txt
rm -rf $out/prof_s5; stamp $out/bench_s5_n1_profiled.json
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  RPVG_AMD_SINGLE_LANE=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_s5_$i -- python /root/repo/bench.py --workload s5 --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_s5_$i.log 2>&1
done
cd /root/repo
timeout 120 python tools/pmc_kernels.py $out/pmc_s5_1 $out/pmc_s5_2 $out/pmc_s5_3 $out/pmc_s5_4 --kernel gibbs,groupConditional,groupLoglik > $out/pmc_s5_conditional_kernels.txt
rm -rf $out/pmc_s5_? $out/pmc_s5_?.log
echo $commit > $out/COMMIT_S5
for f in bench_s3_n1 bench_s5_n1 bench_s5_n1_host_sampler bench_s5_n1_host_threads_4 bench_s5_n1_profiled; do timeout 30 python - <<PY
import json
try:
    d = json.loads(open("$out/$f.json").read())
    print("$f", "ms_per_step", round(d["ms_per_step"], 2), "resident", round(d.get("ms_per_step_resident", 0), 2), "value", round(d["value"] / 1e6, 1), "M/s", "frac", round(d.get("roofline", {}).get("frac", 0), 4))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
head -12 $out/rocprofv3_s5_kernel_stats.csv | cut -c1-100
