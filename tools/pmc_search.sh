#!/bin/bash
# PMC passes over the S3 bench for the search kernels (each pass its own run: counter slots).
cd /tmp; export TMPDIR=/tmp
out=/root/repo/gpurun_out/pmc_search
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU"; do
  i=$((i+1))
  RPVG_AMD_SINGLE_LANE=1 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out.log$i 2>&1
done
cd /root/repo
python tools/pmc_kernels.py $out/p1 $out/p2 $out/p3 $out/p4 $out/p5 --kernel ${1:-pairTile,resolveTable,Search,pairTable}
