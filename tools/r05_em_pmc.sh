#!/bin/bash
# PMC passes over one configs[2] batch at a time (single lane): what the EM kernels issue next to the search and the matrix build
R=/root/repo; out=$R/gpurun_out/r05/em_pmc; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  i=$((i+1))
  RPVG_AMD_SINGLE_LANE=1 RPVG_BENCH_NO_SINGLE=1 RPVG_BENCH_NO_GIBBS_LINE=1 RPVG_BENCH_NO_PIPELINE=1 RPVG_BENCH_NO_HOST_BOUND=1 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1
done
cd $R
python tools/pmc_kernels.py $out/p1 $out/p2 $out/p3 > $out/all_kernels_pmc.txt 2>&1
rm -rf $out/p?
