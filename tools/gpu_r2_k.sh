#!/bin/bash
# Round 2, GPU call K: cost-volume backward row-team schedule A/B (+ prefetch depth), FETCH_SIZE of the winner.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "cost_volume" > gpurun_out/pytest_k.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_k.log | tail -8
for v in "STX_CVB_TEAM=0 STX_CVB_NSET=2" "STX_CVB_TEAM=1 STX_CVB_NSET=2" "STX_CVB_TEAM=1 STX_CVB_NSET=3" "STX_CVB_TEAM=1 STX_CVB_NSET=4"; do
  echo "== cost volume bwd variant [$v]" | tee -a gpurun_out/cvb_ab3.log
  env $v timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume_bwd,cost_volume 2>&1 | grep -E "kernel.*bwd" | tee -a gpurun_out/cvb_ab3.log | cut -c1-150
done
cd /tmp
for v in "STX_CVB_NSET=2" "STX_CVB_NSET=3"; do
for grp in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)_$(echo $v | tr '=' '_')
  env $v timeout 300 rocprofv3 --pmc $grp -d /tmp/pmck_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only cost_volume > /dev/null 2>&1
  echo "-- $v" >> /root/repo/gpurun_out/pmc_cvb3.txt
  python /root/repo/tools/pmc_summary.py /tmp/pmck_$tag cost_volume_bwd >> /root/repo/gpurun_out/pmc_cvb3.txt 2>&1
done
done
cat /root/repo/gpurun_out/pmc_cvb3.txt | cut -c1-150
