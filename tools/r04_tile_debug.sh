#!/bin/bash
# pairTile2Kernel: shapes of the matrices it takes (lane use) and its duration with row classes / loads switched off
# (RPVG_HIP_PAIR_DEBUG bits: 1 count-1 rows, 2 counts 2..8, 4 logarithm rows, 8 single columns, 16 loads; results are wrong then,
# timing only) (gpurun; one lane, rocprofv3 --kernel-trace --stats)
out=/root/repo/gpurun_out/r04/tiledbg; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
[ -n "$NOCLASSES" ] || RPVG_HIP_SEARCH_CLASSES=1 RPVG_AMD_SINGLE_LANE=1 timeout 300 python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "search classes" | head -20 > $out/classes.txt
[ -n "$NOCLASSES" ] || cat $out/classes.txt
for dbg in ${DBG:-0 1 2 4 8 7 15}; do
RPVG_HIP_PAIR_DEBUG=$dbg RPVG_AMD_SINGLE_LANE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$dbg -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/log$dbg 2>&1
python - <<PY
import csv,glob
f=glob.glob("$out/p$dbg/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "pairTile" in r["Name"] or "resolveTable" in r["Name"]: print("debug $dbg", r["Name"].split("(")[0][-30:], "calls", r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1))
PY
rm -rf $out/p$dbg
done
