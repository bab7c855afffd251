#!/bin/bash
# kernel timeline of one configs[2] batch in flight (two host lanes: the engine's default), every launch from MIN_US up; and on one lane
out=gpurun_out/r06/tl${TAG:+_$TAG}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lanes in two one; do
  rm -rf /tmp/prof_tl1
  extra=""; [ $lanes = one ] && extra="RPVG_AMD_SINGLE_LANE=1"
  env $extra RPVG_BENCH_NO_PIPELINE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl1 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/kernel_timeline.py /tmp/prof_tl1 ${MIN_US:-0} > $R/$out/kernel_timeline_s3_one_batch_${lanes}_lanes.txt
  wc -l $R/$out/kernel_timeline_s3_one_batch_${lanes}_lanes.txt
done
