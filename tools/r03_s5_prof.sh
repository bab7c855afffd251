#!/bin/bash
# S5 (configs[4], --use-hap-gibbs) on the GPU box: bench line, kernel statistics and the kernel timeline of one step;
# results under gpurun_out/r03/<tag>.  usage: tools/r03_s5_prof.sh <tag>
tag=${1:-s5}; out=/root/repo/gpurun_out/r03/$tag; mkdir -p $out
cd /root/repo
for i in 1 2; do python bench.py --workload s5 --steps 5 --warmup 2 --no-cpu-baseline 2>$out/bench_$i.err | tail -1 > $out/bench_$i.json; done
python - <<PY
import json
for i in ("1","2"):
    d=json.loads(open("$out/bench_%s.json"%i).read()); k=d["kernels"]
    print(i, "ms_per_step", round(d["ms_per_step"],2), "resident", round(d.get("ms_per_step_resident",0),2), "ll", round(k["loglik_ms_per_step"],2), "build", round(k["build_ms_per_step"],2), "active", d.get("gpu_active_frac"))
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python /root/repo/bench.py --workload s5 --steps 3 --warmup 1 --no-cpu-baseline 2>$out/prof.err | tail -1 > $out/bench_profiled.json
cp $out/prof/*/*kernel_stats.csv $out/kernel_stats.csv
python /root/repo/tools/kernel_timeline.py $out/prof 2 > $out/kernel_timeline.txt
rm -rf $out/prof
head -25 $out/kernel_stats.csv | cut -c1-150
