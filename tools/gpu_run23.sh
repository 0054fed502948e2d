#!/bin/bash
export TMPDIR=/tmp
run() { timeout 120 python tools/kernel_bench.py --iters 10 --only "$1" 2>&1 | grep '"kernel"' | cut -c1-100; }
for i in 1 2; do
echo "=== g8"; run cost_volume | grep bwd
echo "=== generic"; STX_CVB_GENERIC=1 run cost_volume | grep bwd
done
timeout 300 python -m pytest tests/test_kernels.py -m gpu -k cost_volume -q 2>&1 | tail -2
