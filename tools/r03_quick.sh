#!/bin/bash
# quick check on the GPU box: optional tests + S3 bench lines (default host threads and 4) + one-step kernel timeline;
# results under gpurun_out/r03/<tag>.  usage: tools/r03_quick.sh <tag> [pytest args | notests]
tag=${1:-q}; shift
out=/root/repo/gpurun_out/r03/$tag; mkdir -p $out
cd /root/repo
if [ "$1" != "notests" ]; then
  python -m pytest ${@:-tests/test_hip_kernels.py tests/test_hip_collapse.py tests/test_hip_models.py} -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -15 > $out/tests.log
  cat $out/tests.log
fi
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$out/bench_$i.err | tail -1 > $out/bench_$i.json; done
RPVG_AMD_HOST_THREADS=4 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$out/bench_t4.err | tail -1 > $out/bench_t4.json
python - <<PY
import json
for i in ("1","2","3","t4"):
    try:
        d=json.loads(open("$out/bench_%s.json"%i).read()); k=d["kernels"]
        print(i, "ms_per_step", round(d["ms_per_step"],2), "resident", round(d.get("ms_per_step_resident",d["ms_per_step"]),2), "with_h2d", round(d.get("ms_per_step_with_h2d",0),2), "em", round(k["em_sparse_ms_per_step"],2), "ll", round(k["loglik_ms_per_step"],2), "build", round(k["build_ms_per_step"],2), "active", d.get("gpu_active_frac"))
    except Exception as e:
        print(i, "FAILED", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>$out/prof.err | tail -1 > $out/bench_profiled.json
python /root/repo/tools/kernel_timeline.py $out/prof 3 > $out/kernel_timeline.txt
python /root/repo/tools/gpu_gaps.py $out/prof > $out/gpu_gaps.txt 2>&1
rm -rf $out/prof
tail -5 $out/gpu_gaps.txt
