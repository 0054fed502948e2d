#!/bin/bash
# Round 5, call H: counters of the 64->64 L1 implicit-GEMM kernel under both wave grids (STX_CONV_WN = 1 / 2), and a fair cold A/B
# (one process per setting, same kernels first)
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5h
R=$PWD
for wn in 1 2; do
  STX_CONV_WN=$wn timeout 200 python tools/kernel_bench.py --cold --iters 20 --only conv_64_64_L1_fwd,conv_32_64_s2_L0_fwd 2>/dev/null | sed "s/^/WN=$wn cold /" | cut -c1-140
  STX_CONV_WN=$wn timeout 200 python tools/kernel_bench.py --iters 20 --only conv_64_64_L1_fwd,conv_32_64_s2_L0_fwd 2>/dev/null | sed "s/^/WN=$wn warm /" | cut -c1-140
done > ${L}_conv_wn_ab.txt 2>&1; cat ${L}_conv_wn_ab.txt
rm -f ${L}_pmc_igemm_wn.txt
for wn in 1 2; do
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_IFETCH"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && STX_CONV_WN=$wn timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only conv_64_64_L1_fwd > /dev/null 2>&1 )
  echo "== STX_CONV_WN=$wn" >> ${L}_pmc_igemm_wn.txt; python tools/pmc_summary.py /tmp/pmc_x igemm >> ${L}_pmc_igemm_wn.txt 2>&1
done; done
cut -c1-110 ${L}_pmc_igemm_wn.txt
