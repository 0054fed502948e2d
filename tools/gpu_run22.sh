#!/bin/bash
export TMPDIR=/tmp
run() { timeout 120 python tools/kernel_bench.py --iters 10 --only "$1" 2>&1 | grep '"kernel"' | cut -c1-100; }
for i in 1 2; do
echo "=== new 512"; run cost_volume | grep fwd_gwc
echo "=== new 256"; STX_CV_WGS=256 run cost_volume | grep fwd_gwc
echo "=== new 768"; STX_CV_WGS=768 run cost_volume | grep fwd_gwc
echo "=== old";  STX_BENCH_LIB=variants/libstx_old.so run cost_volume | grep fwd_gwc
done
