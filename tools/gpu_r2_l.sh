#!/bin/bash
# Round 2, GPU call L: cost-volume backward after making the loader's loads unconditional (counted waits): schedule x depth.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "cost_volume" > gpurun_out/pytest_l.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_l.log | tail -8
for v in "STX_CVB_TEAM=0 STX_CVB_NSET=2" "STX_CVB_TEAM=0 STX_CVB_NSET=3" "STX_CVB_TEAM=0 STX_CVB_NSET=4" "STX_CVB_TEAM=1 STX_CVB_NSET=2" "STX_CVB_TEAM=1 STX_CVB_NSET=3" "STX_CVB_TEAM=1 STX_CVB_NSET=4"; do
  echo "== cost volume bwd variant [$v]" | tee -a gpurun_out/cvb_ab4.log
  env $v timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume_bwd,cost_volume 2>&1 | grep -E "kernel.*bwd" | tee -a gpurun_out/cvb_ab4.log | cut -c1-150
done
cd /tmp
for v in "STX_CVB_TEAM=0 STX_CVB_NSET=3" "STX_CVB_TEAM=1 STX_CVB_NSET=3"; do
for grp in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $grp | cut -d' ' -f1)_$(echo $v | tr '= ' '__')
  env $v timeout 300 rocprofv3 --pmc $grp -d /tmp/pmcl_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only cost_volume > /dev/null 2>&1
  echo "-- $v" >> /root/repo/gpurun_out/pmc_cvb4.txt
  python /root/repo/tools/pmc_summary.py /tmp/pmcl_$tag cost_volume_bwd >> /root/repo/gpurun_out/pmc_cvb4.txt 2>&1
done
done
cat /root/repo/gpurun_out/pmc_cvb4.txt | cut -c1-150
