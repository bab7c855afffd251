#!/bin/bash
# kernel timeline of one configs[2] batch alone on a single lane (resident batch): every launch with its queue, from 3 us up
out=gpurun_out/r05/tl; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_tl1
env "$@" RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_PIPELINE=1 RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl1 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/kernel_timeline.py /tmp/prof_tl1 ${MIN_US:-3} > $R/$out/kernel_timeline_s3_single_lane.txt
wc -l $R/$out/kernel_timeline_s3_single_lane.txt
