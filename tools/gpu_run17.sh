#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { timeout 120 python tools/kernel_bench.py --iters 5 --only "$1" 2>&1 | grep '"kernel"' | cut -c1-100; }
for v in "" v0 v2 v3; do
  if [ -n "$v" ]; then export STX_BENCH_LIB=variants/libstx_$v.so; else unset STX_BENCH_LIB; fi
  echo "=== lib ${v:-default} pipe";   run conv_32_32_L0_wgrad; run conv_32_64_s2_L0_wgrad
  echo "=== lib ${v:-default} nopipe"; STX_WGRAD_NOPIPE=1 run conv_32_32_L0_wgrad
done
unset STX_BENCH_LIB
for ab in 1 2; do
  echo "=== default ablate=$ab pipe";   STX_WGRAD_ABLATE=$ab run conv_32_32_L0_wgrad
  echo "=== default ablate=$ab nopipe"; STX_WGRAD_ABLATE=$ab STX_WGRAD_NOPIPE=1 run conv_32_32_L0_wgrad
done
