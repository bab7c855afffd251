#!/bin/bash
# SQ counters per kernel of a configs[2] batch (one host lane: every launch a whole batch), four separate rocprofv3 --pmc passes; the summary of the
# tile kernel (instructions per evaluation, VALU busy, LDS bank conflicts) as pmc_search_s3.json
commit=${1:-unknown}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/refresh_pmc; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
pmc() {  # dir, counters, bench args...
  d=$1; c=$2; shift 2
  RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_DROP_IN_LINE=1 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$d -- python $R/bench.py "$@" --no-cpu-baseline > $out/$d.log 2>&1
}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  pmc pmc_search_$i "$set" --steps 2 --warmup 1
done
cd $R
python tools/pmc_kernels.py $out/pmc_search_1 $out/pmc_search_2 $out/pmc_search_3 $out/pmc_search_4 --kernel pairTile,resolveTable,groupsBuild,fillSegments,subsetSelect,subsetMerge,sourceColumns,emRegister,emSparse,collapse,partitionRows,expandGroups,validateRows,segmentsGather,widenNarrow > $out/pmc_s3_throughput_kernels.txt
(cd $out && python $R/tools/pmc_search_summary.py pmc_s3_throughput_kernels.txt 4573105636 pmc_search_s3.json $commit pairTile2Kernel)
rm -rf $out/pmc_search_? $out/*.log
ls $out; head -30 $out/pmc_s3_throughput_kernels.txt
