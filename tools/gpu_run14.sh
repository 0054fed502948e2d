#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmc_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 2 --only cost_volume > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmc_$tag "cost_volume_fwd_row_kernel<8>"
done
