#!/bin/bash
# the default line (configs[2] through the batch pipeline) with more steps, and the EM iteration latencies
out=gpurun_out/r06/bench${TAG:+_$TAG}; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/bench.py --steps ${STEPS:-60} --warmup 8 ${BENCH_ARGS} > $R/$out/bench_s3.json 2> $R/$out/bench_s3.err
python - $R/$out/bench_s3.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "cpu", d.get("cpu_baseline",{}).get("value"))
for k in ("one_batch_in_flight","ms_per_step_with_h2d_serial","h2d_ms_per_batch","h2d_bytes_per_batch","gpu_active_frac","host_cpu_ms_per_step","single_dataset_ms","drop_in"):
    if k in d: print(k, d[k])
print("roofline", d.get("roofline"))
print("em_kernels", json.dumps(d.get("em_kernels"))[:1500])
PY
if [ -n "$EM_LATENCY" ]; then timeout 300 python $R/tools/em_iter_latency.py > $R/$out/em_iteration_latency.txt 2>&1; cat $R/$out/em_iteration_latency.txt | tail -20; fi
