#!/bin/bash
# kernel timeline of one S3 step on the current tree (run on the GPU box through gpurun); result under gpurun_out/r02/<tag>
tag=${1:-tl}
out=/root/repo/gpurun_out/r02/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>$out/prof.err | tail -1 > $out/bench_profiled.json
python /root/repo/tools/kernel_timeline.py $out/prof 3 > $out/kernel_timeline.txt
rm -rf $out/prof
wc -l $out/kernel_timeline.txt
