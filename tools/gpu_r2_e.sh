#!/bin/bash
# Round 2, GPU call E: march v2 (weights in LDS) vs v1, cost-volume builder with batched flush reads, bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "cost_volume or conv3d or dgrad or bn" > gpurun_out/pytest_e.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_e.log | tail -8
for v in "" "STX_CV_WGS=1"; do
  echo "== cost volume variant [$v]" | tee -a gpurun_out/cv_ab5.log
  env $v timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep kernel | tee -a gpurun_out/cv_ab5.log | cut -c1-120
done
for v in "STX_MARCH_V2=1" "STX_MARCH_V2=0" "STX_MARCH_V2=1 STX_MARCH_ABLATE=1" "STX_MARCH_V2=1 STX_MARCH_ABLATE=2"; do
  echo "== march variant [$v]" | tee -a gpurun_out/march_ab.log
  env $v timeout 300 python tools/kernel_bench.py --iters 10 --only conv_32_32_L0_fwd 2>&1 | grep kernel | tee -a gpurun_out/march_ab.log | cut -c1-120
done
for v in "STX_MARCH_V2=1" "STX_MARCH_V2=0"; do
  echo "== bench [$v]" | tee -a gpurun_out/bench_e.log
  env $v timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/bench_e.log | cut -c1-330
done
