#!/bin/bash
# the driver's command, python bench.py, as is: wall time and the fields of the line that are new in round 6
out=gpurun_out/r06/default${TAG:+_$TAG}; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
start=$(date +%s.%N)
python $R/bench.py > $R/$out/bench_default.json 2> $R/$out/bench_default.err
end=$(date +%s.%N)
echo "wall seconds $(echo "$end - $start" | bc)"
python - $R/$out/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value","ms_per_step","vs_baseline","vs_cpu_baseline","single_dataset_ms","h2d_bytes_per_batch","h2d_ms_per_batch","pipeline_estimates_equal_single_engine"):
    print(k, d.get(k))
print("drop_in", d.get("drop_in"))
print("single_dataset", d.get("single_dataset"))
print("cpu", d.get("cpu_baseline"))
print("gibbs", {k:v for k,v in d.get("roofline_gibbs",{}).items() if k in ("value","batch_ms_per_step","host_cpu_ms_per_step","gpu_active_frac","error")})
PY
