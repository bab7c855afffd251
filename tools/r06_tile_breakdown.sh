#!/bin/bash
# pairTile2Kernel on an EXPERIMENTS=1 build of rpvg_amd/csrc (make clean all EXPERIMENTS=1): the shapes of the matrices it takes and the use
# of its lanes (RPVG_HIP_SEARCH_CLASSES), and its duration with classes of rows / the loads switched off (RPVG_HIP_PAIR_DEBUG bits: 1 count-1
# rows, 2 counts 2..8, 4 logarithm rows, 8 single columns, 16 loads, 32 the epilogue, 64 everything behind the prologue, 128 everything; results are
# wrong then, timing only).  One lane, rocprofv3 --kernel-trace --stats.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06/tiledbg; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
RPVG_HIP_SEARCH_CLASSES=1 RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_PIPELINE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 300 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "search classes" | head -9 > $out/pair_tile_breakdown.txt
for dbg in ${DBG:-0 1 2 4 8 15 16 31 63 95 159}; do
RPVG_HIP_PAIR_DEBUG=$dbg RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_PIPELINE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$dbg -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/log$dbg 2>&1
python - <<PY >> $out/pair_tile_breakdown.txt
import csv,glob
f=glob.glob("$out/p$dbg/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "pairTile" in r["Name"]: print("RPVG_HIP_PAIR_DEBUG=$dbg", r["Name"].split("(")[0][-30:], "calls", r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1))
PY
rm -rf $out/p$dbg $out/log$dbg
done
cat $out/pair_tile_breakdown.txt
