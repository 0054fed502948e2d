#!/bin/bash
# Round 4, call U: counters of the stride-2 march weight-gradient kernel (32->64 L0 and 64->128 L1 launches)
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4u
R=$PWD
rm -f ${L}_pmc_wgrad_march_s2.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only conv_32_64_s2_L0_wgrad,conv_64_128_s2_L1_wgrad > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x wgrad_march_s2 >> ${L}_pmc_wgrad_march_s2.txt 2>&1
done
cut -c1-110 ${L}_pmc_wgrad_march_s2.txt
