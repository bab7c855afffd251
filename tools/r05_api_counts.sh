#!/bin/bash
# HIP API calls per pipelined configs[2] batch (rocprofv3 --hip-trace --stats): which stream commands a batch issues besides its kernels
out=gpurun_out/r05/api; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_api
STEPS=${STEPS:-40}
RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/prof_api -- python $R/bench.py --steps $STEPS --warmup 4 --no-cpu-baseline > /dev/null 2>&1
f=$(ls /tmp/prof_api/*/*hip_api_stats.csv 2>/dev/null | tail -1)
[ -z "$f" ] && f=$(ls /tmp/prof_api/*/*stats*.csv | head -1)
cp $f $R/$out/hip_api_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -int(r["Calls"]))
for r in rows[:25]:
    print(f'{r["Name"]:44s} calls {int(r["Calls"]):8d} total ms {float(r["TotalDurationNs"])/1e6:10.1f} avg us {float(r["AverageNs"])/1e3:8.1f}')
PY
