#!/bin/bash
# kernel timelines of one configs[2] step: two lanes (default) and a single lane (no contention between lanes)
out=gpurun_out/r04/tl; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl2 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/kernel_timeline.py /tmp/prof_tl2 3 > $R/$out/kernel_timeline_s3_two_lanes.txt
RPVG_AMD_SINGLE_LANE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl1 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/kernel_timeline.py /tmp/prof_tl1 3 > $R/$out/kernel_timeline_s3_single_lane.txt
wc -l $R/$out/*.txt
