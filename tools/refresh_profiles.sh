#!/bin/bash
# Re-measures the bench lines and the S3 / rows kernel profiles into gpurun_out/refresh/ (copy what is kept to profiles/).
out=/root/repo/gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd /root/repo
python bench.py --steps 20 --warmup 3 2>$out/bench_s3.err | tail -1 > $out/bench_s3_n1.json
python bench.py --workload e2e --steps 10 --warmup 2 2>$out/bench_e2e.err | tail -1 > $out/bench_e2e_n1.json
python bench.py --workload rows --steps 10 --warmup 2 2>$out/bench_rows.err | tail -1 > $out/bench_rows_n1.json
python bench.py --workload s5 --steps 5 --warmup 1 2>$out/bench_s5.err | tail -1 > $out/bench_s5_n1.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_s3 -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>$out/prof_s3.err | tail -1 > $out/bench_s3_n1_profiled.json
cp $out/prof_s3/*/*kernel_stats.csv $out/rocprofv3_s3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_rows -- python /root/repo/bench.py --workload rows --steps 10 --warmup 2 --no-cpu-baseline 2>$out/prof_rows.err | tail -1 > $out/bench_rows_n1_profiled.json
cp $out/prof_rows/*/*kernel_stats.csv $out/rocprofv3_rows_kernel_stats.csv
rm -rf $out/prof_s3 $out/prof_rows
cd /root/repo
for f in bench_s3_n1 bench_e2e_n1 bench_rows_n1 bench_s5_n1 bench_s3_n1_profiled bench_rows_n1_profiled; do python - <<PY
import json
d=json.loads(open("$out/$f.json").read())
print("$f", round(d["ms_per_step"],2), round(d["value"]/1e6,1), d.get("cpu_baseline",{}).get("value"))
PY
done
