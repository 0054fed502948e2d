#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3bal2
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -2 ) | cut -c1-200
for v in 1 0 1 0; do STX_WGRAD_BAL=$v timeout 200 python tools/kernel_bench.py --iters 30 --only wgrad 2>&1 | grep '"kernel"' | grep -v "c1\|1x1" | sed "s/^/bal=$v /" | tee -a ${L}_wgrad.txt | cut -c1-120; done
