#!/bin/bash
# A/B of the lanes' shares of the batch on the S3 bench (gpurun); results under gpurun_out/r02/<tag>
tag=${1:-shares}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
for rep in 1 2 3; do for sh in 50,50 60,40 70,30 40,60; do
RPVG_AMD_LANE_SHARES=$sh python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_${sh}_$rep.json
python -c "
import json; d=json.loads(open('$out/bench_${sh}_$rep.json').read()); print('shares $sh', round(d['ms_per_step'],2), 'with_h2d', round(d.get('ms_per_step_with_h2d',0),2))"
done; done
