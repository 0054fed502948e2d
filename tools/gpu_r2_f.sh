#!/bin/bash
# Round 2, GPU call F: march v2 with mid-step staging + 64->32 / 32->64 as 32x32 slices; kernel table of the conv shapes; bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "conv3d or dgrad or bn or deconv" > gpurun_out/pytest_f.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_f.log | tail -8
for v in "STX_MARCH_V2=1" "STX_MARCH_V2=0"; do
  echo "== conv variant [$v]" | tee -a gpurun_out/conv_ab_f.log
  env $v timeout 300 python tools/kernel_bench.py --iters 10 --only _L0_fwd --skip-wgrad 2>&1 | grep -E "conv_" | tee -a gpurun_out/conv_ab_f.log | cut -c1-120
done
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_f.log | cut -c1-330
timeout 900 python -m pytest tests/test_models.py -m gpu -q -p no:cacheprovider -k "train_parity or full_size_train or eval_parity" > gpurun_out/pytest_f2.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_f2.log | tail -8
