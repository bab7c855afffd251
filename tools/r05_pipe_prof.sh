#!/bin/bash
# rocprofv3 kernel statistics of the pipelined configs[2] bench (several batches in flight), round 5
out=gpurun_out/r05/prof; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RPVG_BENCH_NO_SINGLE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pipe -- python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline > $R/$out/bench_profiled.json 2>/dev/null
f=$(ls /tmp/prof_pipe/*/*kernel_stats.csv | tail -1)
cp $f $R/$out/rocprofv3_s3_pipeline_kernel_stats.csv
python $R/tools/kernel_stats_table.py $f 5 | head -60
python - <<PY
import json
d=json.load(open("$R/$out/bench_profiled.json"))
print("profiled ms_per_step", d["ms_per_step"], "steps", d["steps"])
PY
