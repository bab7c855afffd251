#!/bin/bash
# average duration of the kernels whose names contain $1 (comma-separated) in a one-lane run of the configs[2] bench (gpurun)
out=/root/repo/gpurun_out/r04/ktime; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
RPVG_AMD_SINGLE_LANE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$out/p/*/*kernel_stats.csv")[0]
want="$1".split(",")
for r in csv.DictReader(open(f)):
    if any(w in r["Name"] for w in want): print(r["Name"].split("(")[0][-40:], "calls", r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1))
PY
rm -rf $out/p
