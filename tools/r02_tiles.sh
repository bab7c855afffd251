#!/bin/bash
# pair-tile search: tests + A/B on the S3 bench (gpurun); results under gpurun_out/r02/<tag>
tag=${1:-tiles}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_models.py tests/test_hip_collapse.py tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15 > $out/tests.log
cat $out/tests.log
for rep in 1 2 3; do for t in 0 1; do
RPVG_HIP_PAIR_TILES=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$out/bench_t${t}_$rep.err | tail -1 > $out/bench_t${t}_$rep.json
python -c "
import json; d=json.loads(open('$out/bench_t${t}_$rep.json').read()); print('tiles $t', round(d['ms_per_step'],2), 'with_h2d', round(d.get('ms_per_step_with_h2d',0),2), 'loglik', round(d['kernels']['loglik_ms_per_step'],2), 'evals', d['kernels']['loglik_evals_per_step'], d['mass_conserved'])"
done; done
