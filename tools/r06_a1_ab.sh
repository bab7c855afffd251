#!/bin/bash
# A/B of the call combiner's settings on the a1 line (team of 64): each variant = environment assignments, 6 steps
out=gpurun_out/r06/a1_ab${TAG:+_$TAG}; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
i=0
while IFS= read -r variant; do
  [ -z "$variant" ] && continue
  i=$((i+1))
  env $variant timeout ${PER_RUN_TIMEOUT:-150} python $R/bench.py --workload a1 --team ${TEAM:-64} --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline > $R/$out/v$i.json 2> $R/$out/v$i.err
  python - "$variant" $R/$out/v$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f'{sys.argv[1]:70s} {d["ms_per_step"]:8.1f} ms  {d["value"]/1e6:7.1f} M/s  equal={d["estimates_equal_estimate_batch"]}  {d["ms_per_step_in_order"]}')
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done <<< "$VARIANTS"
