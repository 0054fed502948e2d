#!/bin/bash
# Round 2, GPU call R: PMC of the convolution kernels that sit at 0.45-0.62 of the fp32-MFMA peak (what stalls them?).
mkdir -p gpurun_out
export TMPDIR=/tmp
K="deconv_64_32_L1_fwd,deconv_128_64_L2_fwd,conv_32_64_s2_L0_fwd,conv_64_64_L1_fwd,conv_64_128_s2_L1_fwd"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  ( cd /tmp && timeout 60 rocprofv3 --pmc $grp -d /tmp/pmcr_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --skip-wgrad --only $K > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmcr_$tag conv >> gpurun_out/pmc_convs_r.txt 2>&1
done
cut -c1-140 gpurun_out/pmc_convs_r.txt
