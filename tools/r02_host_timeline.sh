#!/bin/bash
# host phase timeline of one S3 step (run on the GPU box through gpurun); result under gpurun_out/r02/<tag>
tag=${1:-htl}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
RPVG_AMD_TIMELINE=1 RPVG_AMD_TRACE=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/timeline.err
python tools/timeline_step.py $out/timeline.err 3 > $out/host_timeline.txt
python tools/trace_summary.py < $out/timeline.err > $out/trace_summary.txt
rm -f $out/timeline.err
wc -l $out/host_timeline.txt
