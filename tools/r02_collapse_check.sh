#!/bin/bash
# collapse tests + S3 bench + kernel profile (run on the GPU box through gpurun); results under gpurun_out/r02/<tag>
tag=${1:-x}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
python -m pytest tests/test_hip_collapse.py -x -q -m gpu 2>&1 | tail -25 > $out/collapse_tests.log
RPVG_AMD_TRACE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "row collapse" | tail -2 > $out/collapse_trace.log
python bench.py --no-cpu-baseline > $out/bench_s3.json 2> $out/bench_s3.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>$out/prof.err | tail -1 > $out/bench_s3_profiled.json
cp $out/prof/*/*kernel_stats.csv $out/rocprofv3_s3_kernel_stats.csv; rm -rf $out/prof
RPVG_AMD_SINGLE_LANE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof1 -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>$out/prof1.err | tail -1 > $out/bench_s3_profiled_1lane.json
cp $out/prof1/*/*kernel_stats.csv $out/rocprofv3_s3_kernel_stats_1lane.csv; rm -rf $out/prof1
cat $out/collapse_tests.log $out/collapse_trace.log
python - <<PY
import json, csv
d=json.loads(open("$out/bench_s3.json").read().strip().splitlines()[-1]); print("ms_per_step", d["ms_per_step"], d["kernels"])
for r in csv.DictReader(open("$out/rocprofv3_s3_kernel_stats.csv")):
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][-50:]
    if float(r['TotalDurationNs'])/1e6 > 1.0: print(f"{n:52s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.1f} ms")
print("---- one lane")
for r in csv.DictReader(open("$out/rocprofv3_s3_kernel_stats_1lane.csv")):
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][-50:]
    if 'collapse' in n or 'radix' in r['Name'] or 'Build' in n or 'partition' in n: print(f"{n:52s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.1f} ms")
PY
