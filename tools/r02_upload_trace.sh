#!/bin/bash
# host phases of the batch upload (bench's with-H2D leg) (gpurun); results under gpurun_out/r02/<tag>
tag=${1:-upl}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
RPVG_AMD_TRACE=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2> $out/trace.err | tail -1 > $out/bench.json
python tools/trace_summary.py < $out/trace.err | grep -E "batch_upload|device batch" > $out/upload_phases.txt
rm -f $out/trace.err
cat $out/upload_phases.txt
python -c "
import json; d=json.loads(open('$out/bench.json').read()); print({k:d[k] for k in d if 'h2d' in k and k!='h2d_note'})"
