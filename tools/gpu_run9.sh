#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/kernel_bench.py --iters 5 --only _L0_fwd 2>&1 | grep -E "conv_(32_32|64_32)"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/pmc_a -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 2 --only conv_32_32_L0_fwd > /dev/null 2>&1
python /root/repo/tools/pmc_summary.py /tmp/pmc_a conv3d
