#!/bin/bash
# Round 5, call R: counters of the BatchNorm-backward form of the march weight gradient (conv3d_wgrad_march_kernel<true>) inside
# the train step (VERDICT r4 "evidence hygiene": no PMC of that form existed) -- four separate passes over a 2-step run
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5r
R=$PWD
rm -f ${L}_pmc_wgrad_march_bn.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 400 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x "wgrad_march_kernel<true>" >> ${L}_pmc_wgrad_march_bn.txt 2>&1
done
cut -c1-110 ${L}_pmc_wgrad_march_bn.txt
