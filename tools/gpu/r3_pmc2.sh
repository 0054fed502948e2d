#!/bin/bash
# Round 3: dynamic instruction mix (per-class instruction counts) of the weight-gradient, march and stride-2 kernels.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3pmc2
R=$PWD
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only conv_32_32_L0_wgrad,conv_32_32_L0_fwd,conv_32_64_s2_L0_fwd,conv_64_64_L1_fwd,deconv_64_32_L1_fwd > /tmp/pmc_x.log 2>&1; tail -2 /tmp/pmc_x.log | cut -c1-200 )
  python tools/pmc_summary.py /tmp/pmc_x conv >> ${L}.txt 2>&1
done
grep -v pack_kernel ${L}.txt | cut -c1-110 | head -90
