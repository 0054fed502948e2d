#!/bin/bash
# Round 5, call J: the transposed kernels after keeping the 64-channel instantiation on per-lane wave indices (A/B against the
# previous build), counters of the stride-2 32->64 and the transposed 64->32 kernels on the new build, the headline line.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5j
R=$PWD
B=$PWD/stereo_toolbox_amd/lib/libstx_hip_before.so
for rep in 1 2; do
  STX_BENCH_LIB=$B timeout 200 python tools/kernel_bench.py --cold --iters 20 --only deconv 2>/dev/null | sed "s/^/before$rep /" | cut -c1-150
  timeout 200 python tools/kernel_bench.py --cold --iters 20 --only deconv 2>/dev/null | sed "s/^/after$rep  /" | cut -c1-150
done > ${L}_deconv_ab.txt 2>&1; sort -k3,3 -s ${L}_deconv_ab.txt | cut -c1-150
rm -f ${L}_pmc_s2_deconv.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only conv_32_64_s2_L0_fwd,deconv_64_32_L1_fwd > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x igemm >> ${L}_pmc_s2_deconv.txt 2>&1
done
cut -c1-110 ${L}_pmc_s2_deconv.txt
timeout 300 python bench.py --config gwc_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_gwc.json"))
print("gwc_train", d["value"], d["ms_per_step"], d.get("hot_path_ms"), d.get("feature_cnn_ms"))
EOF2
