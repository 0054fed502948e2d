#!/bin/bash
# Round 2, GPU call S: transposed-conv variants (occupancy hint: 2 waves per SIMD; tap list depth; K chunk).
mkdir -p gpurun_out
export TMPDIR=/tmp
( for v in "" "STX_DECONV_PIPE=0" "STX_DECONV_PIPE=2" "STX_DECONV_CK=16" "STX_DECONV_CK=16 STX_DECONV_PIPE=2"; do env $v timeout 100 python -m pytest tests -m gpu -q -p no:cacheprovider -k "deconv or dgrad" 2>&1 | tail -1; done ) > gpurun_out/pytest_gpu_s.log 2>&1
cat gpurun_out/pytest_gpu_s.log
timeout 120 python tools/kernel_bench.py --iters 20 --ab --only deconv > gpurun_out/kernel_bench_s.log 2>&1; grep -E '"kernel"|"ab"' gpurun_out/kernel_bench_s.log | grep -E "deconv|transposed" > gpurun_out/kernel_bench_s.jsonl; cut -c1-140 gpurun_out/kernel_bench_s.jsonl
