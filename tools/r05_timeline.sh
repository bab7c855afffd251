#!/bin/bash
# kernel timeline + GPU gaps of the configs[2] steps (two lanes, uploads in the loop), round 5
out=gpurun_out/r05/tl; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl2 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/kernel_timeline.py /tmp/prof_tl2 15 > $R/$out/kernel_timeline_s3_two_lanes.txt
python $R/tools/gpu_gaps.py /tmp/prof_tl2 60 > $R/$out/gpu_gaps_s3.txt
cat $R/$out/gpu_gaps_s3.txt
cat $R/$out/kernel_timeline_s3_two_lanes.txt | head -150
