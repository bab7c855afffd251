#!/bin/bash
# the whole -m gpu suite + the default bench line (gpurun); results under gpurun_out/r02/<tag>
tag=${1:-full}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15 > $out/tests.log
cat $out/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $out/bench_s3.json 2> $out/bench_s3.err
python -c "
import json; d=json.loads(open('$out/bench_s3.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'with_h2d', d.get('ms_per_step_with_h2d'), d['roofline_search'], d['cpu_baseline']['value'])"
