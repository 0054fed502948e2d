#!/bin/bash
export TMPDIR=/tmp
run() { timeout 120 python tools/kernel_bench.py --iters 10 --only "$1" 2>&1 | grep '"kernel"' | cut -c1-100; }
for ab in 0 0 1 6 8; do echo "=== ablate $ab"; STX_CVL_ABLATE=$ab run cost_volume | grep fwd_gwc; done
echo "=== rowkernel"; STX_CV_NO_G8=1 run cost_volume | grep fwd_gwc
timeout 300 python -m pytest tests/test_kernels.py -m gpu -k cost_volume -q 2>&1 | tail -2
