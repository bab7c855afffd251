#!/bin/bash
# A/B of environment settings on the S3 bench (gpurun): tools/r02_env_ab.sh <tag> "<env assignments>" ...
tag=$1; shift
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
for rep in 1 2 3; do
  i=0
  for envs in "$@"; do
    i=$((i+1))
    env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_${i}_$rep.json
    python -c "
import json; d=json.loads(open('$out/bench_${i}_$rep.json').read()); print('%-40s' % '$envs', round(d['ms_per_step'],2), 'with_h2d', round(d.get('ms_per_step_with_h2d',0),2), 'upload', round(d.get('h2d_ms_per_batch',0),2))"
  done
done
