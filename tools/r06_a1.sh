#!/bin/bash
# The drop-in path (PathEstimator::estimate() per cluster from an OpenMP team) on configs[2]: phase totals, host timeline of the
# batches, kernel trace of one pass.  TEAM=64 by default.
out=gpurun_out/r06/a1${TAG:+_$TAG}; mkdir -p $out
R=$GRAFT_REPO_ROOT
TEAM=${TEAM:-64}
cd /tmp; export TMPDIR=/tmp
# 1. the line itself
python $R/bench.py --workload a1 --team $TEAM --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline > $R/$out/bench_a1_team_$TEAM.json 2> $R/$out/bench_a1.err
tail -c 1500 $R/$out/bench_a1_team_$TEAM.json
# 2. phase totals (summed over threads) per pass
RPVG_AMD_TRACE=1 python $R/bench.py --workload a1 --team $TEAM --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/$out/phase_totals_team_$TEAM.txt
grep -c "trace" $R/$out/phase_totals_team_$TEAM.txt
# 3. host timeline of one pass
RPVG_AMD_TIMELINE=1 python $R/bench.py --workload a1 --team $TEAM --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/timeline.txt
grep "^\[timeline\]" /tmp/timeline.txt | grep -v "flatten the cluster" > $R/$out/host_timeline_team_$TEAM.txt
grep -c "flatten the cluster" /tmp/timeline.txt
wc -l $R/$out/host_timeline_team_$TEAM.txt
# 4. kernel trace of a short run
rm -rf /tmp/prof_a1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a1 -- python $R/bench.py --workload a1 --team $TEAM --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(ls /tmp/prof_a1/*/*kernel_stats.csv | tail -1); cp $f $R/$out/rocprofv3_a1_kernel_stats.csv
t=$(ls /tmp/prof_a1/*/*kernel_trace.csv | tail -1); gzip -c $t > $R/$out/a1_kernel_trace.csv.gz
ls -la $R/$out
