#!/bin/bash
# host phases of the S5 step (configs[4], --use-hap-gibbs) (gpurun); results under gpurun_out/r02/<tag>
tag=${1:-s5trace}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
RPVG_AMD_TRACE=1 python bench.py --workload s5 --steps 3 --warmup 1 --no-cpu-baseline 2> $out/trace.err | tail -1 > $out/bench.json
python tools/trace_summary.py < $out/trace.err > $out/trace_summary.txt
rm -f $out/trace.err
head -40 $out/trace_summary.txt
