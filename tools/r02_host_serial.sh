#!/bin/bash
# host phases of the S3 step with ONE host thread and one lane: wall time = CPU time of each phase (gpurun)
tag=${1:-hostserial}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /root/repo
RPVG_AMD_HOST_THREADS=1 RPVG_AMD_SINGLE_LANE=1 RPVG_AMD_TRACE=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2> $out/trace.err | tail -1 > $out/bench.json
python tools/trace_summary.py < $out/trace.err > $out/trace_summary.txt
rm -f $out/trace.err
cat $out/trace_summary.txt
