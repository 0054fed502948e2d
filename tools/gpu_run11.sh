#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/kernel_bench.py --iters 5 --only cost_volume 2>&1 | grep -v amdgpu
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmc_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 2 --only cost_volume > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmc_$tag cost_volume_fwd
done
