#!/bin/bash
# quick check on the GPU box: selected tests + three S3 bench lines; results under gpurun_out/r02/<tag>
tag=${1:-q}; shift
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
python -m pytest ${@:-tests/test_hip_kernels.py tests/test_hip_collapse.py tests/test_hip_models.py} -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -15 > $out/tests.log
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$out/bench_$i.err | tail -1 > $out/bench_$i.json; done
cat $out/tests.log
python - <<PY
import json
for i in (1,2,3):
    d=json.loads(open("$out/bench_%d.json"%i).read()); print("ms_per_step", round(d["ms_per_step"],2), "with_h2d", round(d.get("ms_per_step_with_h2d",0),2), d["kernels"]["em_sparse_ms_per_step"], d["kernels"]["loglik_ms_per_step"], d["kernels"]["build_ms_per_step"])
PY
