#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "--- pipe" ; timeout 300 python tools/kernel_bench.py --iters 5 --only wgrad 2>&1 | grep -v c1 | cut -c1-120
echo "--- nopipe"; STX_WGRAD_NOPIPE=1 timeout 300 python tools/kernel_bench.py --iters 5 --only wgrad 2>&1 | grep -v c1 | cut -c1-120
