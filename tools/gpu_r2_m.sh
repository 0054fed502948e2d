#!/bin/bash
# Round 2, GPU call M: ablation of the cost-volume backward roles (which one sets the step time).
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "STX_CVB_ABLATE=0" "STX_CVB_ABLATE=1" "STX_CVB_ABLATE=2" "STX_CVB_ABLATE=4" "STX_CVB_ABLATE=8" "STX_CVB_ABLATE=6" "STX_CVB_ABLATE=14" "STX_CVB_ABLATE=15" "STX_CVB_ABLATE=0 STX_CVB_TEAM=0" "STX_CVB_ABLATE=15 STX_CVB_TEAM=0"; do
  echo "== [$v]" | tee -a gpurun_out/cvb_ablate.log
  env $v timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume_bwd,cost_volume 2>&1 | grep -E "kernel.*bwd" | tee -a gpurun_out/cvb_ablate.log | cut -c1-150
done
