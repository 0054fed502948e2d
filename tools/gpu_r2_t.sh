#!/bin/bash
# Round 2, GPU call T: K-chunk size (= resident workgroups per CU) of the transposed conv and of the stride-1 L1/L2 convs.
mkdir -p gpurun_out
export TMPDIR=/tmp
( for v in "" "STX_DECONV_CK=8" "STX_DECONV_CK=32"; do env $v timeout 100 python -m pytest tests -m gpu -q -p no:cacheprovider -k "deconv or dgrad" 2>&1 | tail -1; done ) > gpurun_out/pytest_gpu_t.log 2>&1
cat gpurun_out/pytest_gpu_t.log
timeout 120 python tools/kernel_bench.py --iters 20 --ab --only deconv > gpurun_out/kernel_bench_t.log 2>&1; grep -E '"kernel"|"ab"' gpurun_out/kernel_bench_t.log | grep -E "deconv|transposed" > gpurun_out/kernel_bench_t.jsonl; cut -c1-140 gpurun_out/kernel_bench_t.jsonl
for v in "X=0" "STX_CONV_CK=16" "STX_CONV_CK=8" "STX_CONV_PIPE=0" "STX_CONV_PIPE=0 STX_CONV_CK=16" "STX_CONV_PIPE=2" "STX_CONV_PIPE=2 STX_CONV_CK=16"; do
  echo "== [$v]" | tee -a gpurun_out/conv_ck_t.txt
  env $v timeout 60 python tools/kernel_bench.py --iters 20 --skip-wgrad --only conv_64_64_L1_fwd,conv_128_128_L2_fwd,conv_64_128_s2_L1_fwd,conv_32_64_s2_L0_fwd 2>&1 | grep kernel | tee -a gpurun_out/conv_ck_t.txt | cut -c1-110
done
