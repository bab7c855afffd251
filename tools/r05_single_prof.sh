#!/bin/bash
# rocprofv3 kernel statistics of configs[2] batches one at a time on a single lane: every kernel's time standing alone
out=gpurun_out/r05/prof; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_single
env "$@" RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_PIPELINE=1 RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_single -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/$out/bench_single_profiled.json 2>/dev/null
f=$(ls /tmp/prof_single/*/*kernel_stats.csv | tail -1)
cp $f $R/$out/rocprofv3_s3_single_lane_kernel_stats.csv
python $R/tools/kernel_stats_table.py $f 5 | head -${TOP:-40}
