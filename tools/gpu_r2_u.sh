#!/bin/bash
# Round 2, GPU call U: occupancy hints + 8-channel chunks as the new defaults of the implicit-GEMM convolutions.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 150 python -m pytest tests -m gpu -q -p no:cacheprovider -k "conv3d or deconv or dgrad or layer_shapes or gwcnet_gc_train" 2>&1 | tail -3 ) > gpurun_out/pytest_gpu_u.log 2>&1
cat gpurun_out/pytest_gpu_u.log
timeout 120 python tools/kernel_bench.py --iters 20 --ab --skip-wgrad --only deconv,conv_64_64_L1_fwd,conv_128_128_L2_fwd,conv_64_128_s2_L1_fwd,conv_32_64_s2_L0_fwd,conv1x1 > gpurun_out/kernel_bench_u.log 2>&1; grep -E '"kernel"|"ab"' gpurun_out/kernel_bench_u.log | grep -E "conv|transposed" > gpurun_out/kernel_bench_u.jsonl; cut -c1-140 gpurun_out/kernel_bench_u.jsonl
for v in "STX_CONV_CK=16" "STX_CONV_CK=32" "STX_CONV_PIPE=0"; do
  echo "== [$v]" | tee -a gpurun_out/conv_ck_u.txt
  env $v timeout 60 python tools/kernel_bench.py --iters 20 --skip-wgrad --only conv_64_64_L1_fwd,conv_128_128_L2_fwd,conv_64_128_s2_L1_fwd,conv_32_64_s2_L0_fwd 2>&1 | grep kernel | tee -a gpurun_out/conv_ck_u.txt | cut -c1-110
done
