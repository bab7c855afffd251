#!/bin/bash
# A/B of the number of host lanes per engine on the S3 bench (gpurun); results under gpurun_out/r02/<tag>
tag=${1:-lanes}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
for rep in 1 2; do for n in 2 3 4; do
RPVG_AMD_LANES=$n python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_l${n}_$rep.json
python -c "
import json; d=json.loads(open('$out/bench_l${n}_$rep.json').read()); print('lanes $n', round(d['ms_per_step'],2), 'with_h2d', round(d.get('ms_per_step_with_h2d',0),2))"
done; done
