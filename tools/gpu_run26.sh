#!/bin/bash
export TMPDIR=/tmp
run() { timeout 120 python tools/kernel_bench.py --iters 10 --only "$1" 2>&1 | grep '"kernel"' | cut -c1-100; }
for i in 1 2 3; do echo "=== 2chain"; run cost_volume | grep fwd_gwc; echo "=== 1chain"; STX_BENCH_LIB=variants/libstx_old.so run cost_volume | grep fwd_gwc; done
