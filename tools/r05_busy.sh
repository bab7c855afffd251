#!/bin/bash
# the pipeline's steady state in a kernel trace: share of time with a kernel running, kernels running at once, a stretch of its timeline (round 5)
out=gpurun_out/r05/busy; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_busy -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline > $R/$out/bench.json 2>/dev/null
python $R/tools/gpu_busy_union.py /tmp/prof_busy 16 | tee $R/$out/gpu_busy_s3_pipeline.txt
python $R/tools/r05_pipe_timeline.py /tmp/prof_busy 14 25 > $R/$out/kernel_timeline_s3_pipeline.txt
python - <<PY
import json
d=json.load(open("$R/$out/bench.json")); print("traced run: ms_per_step", round(d["ms_per_step"],2), "gpu_active_frac (spans)", round(d["gpu_active_frac"],3))
PY
