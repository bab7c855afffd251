#!/bin/bash
# PMC passes over the lone-problem EM microbenchmark (tools/em_iter_latency.py): per-iteration cycles, instructions, LDS waits
out=/root/repo/gpurun_out/r03/em_pmc; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python /root/repo/tools/em_iter_latency.py 20000 > $out/p$i.log 2>&1
done
cd /root/repo
python tools/pmc_kernels.py $out/p1 $out/p2 $out/p3 $out/p4 --kernel emRegisterKernel,emSparseKernel > $out/em_pmc.txt 2>&1
rm -rf $out/p?
head -100 $out/em_pmc.txt
