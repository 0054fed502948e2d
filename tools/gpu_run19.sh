#!/bin/bash
export TMPDIR=/tmp
run() { timeout 120 python tools/kernel_bench.py --iters 5 --only "$1" 2>&1 | grep '"kernel"' | cut -c1-100; }
for i in 1 2; do
echo "=== wholeK MW16"; run conv_32_32_L0_fwd
echo "=== wholeK MW32"; STX_MARCH_MW32=1 run conv_32_32_L0_fwd
echo "=== ksplit MW16"; STX_MARCH_KSPLIT=1 run conv_32_32_L0_fwd
echo "=== old";  STX_BENCH_LIB=variants/libstx_old.so run conv_32_32_L0_fwd
done
