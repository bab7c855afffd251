#!/bin/bash
# HIP API calls of the drop-in path (rocprofv3 --hip-trace --stats): which runtime calls a pass of 5 000 estimate() calls issues and what they cost the leaders
out=gpurun_out/r06/a1_api; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_api
timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/prof_api -- python $R/bench.py --workload a1 --team 64 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(ls /tmp/prof_api/*/*hip_api_stats.csv 2>/dev/null | tail -1)
[ -z "$f" ] && f=$(ls /tmp/prof_api/*/*stats*.csv | head -1)
cp $f $R/$out/hip_api_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total API ms", tot/1e6, "(5 passes incl. warmup and the checks)")
for r in rows[:22]:
    print(f'{r["Name"]:44s} calls {int(r["Calls"]):8d} total ms {float(r["TotalDurationNs"])/1e6:10.1f} avg us {float(r["AverageNs"])/1e3:8.1f}')
PY
