#!/bin/bash
# A/B of an environment setting on the default line's EM figures: per variant ms per launch and us per iteration of the slowest problem
out=gpurun_out/r06/em_ab${TAG:+_$TAG}; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
i=0
while IFS= read -r variant; do
  [ -z "$variant" ] && continue
  i=$((i+1))
  env $variant RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 RPVG_BENCH_NO_SINGLE_DATASET=1 timeout 300 python $R/bench.py --steps ${STEPS:-40} --warmup 6 --no-cpu-baseline > $R/$out/v$i.json 2> $R/$out/v$i.err
  python - "$variant" $R/$out/v$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    one=d["one_batch_in_flight"]
    print(f'{sys.argv[1]:40s} pipeline {d["ms_per_step"]:.2f} ms  one batch {one["ms_per_step_with_h2d"]:.2f} (median {one["ms_per_step_spread"]["median"]:.2f})')
    for k,v in d["em_kernels"].items():
        print(f'    {k:28s} {v["ms_per_launch"]:.3f} ms  slowest {v["slowest_problem_iterations"]:.0f} its  {v["us_per_iteration_of_slowest"]:.3f} us/it')
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done <<< "$VARIANTS"
