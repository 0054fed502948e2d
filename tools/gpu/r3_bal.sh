#!/bin/bash
# Round 3: balanced tap split of the 3x3x3 weight gradient (STX_WGRAD_BAL).
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3bal
( timeout 600 python -m pytest tests/test_kernels.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider -k "wgrad or block or reproduc" 2>&1 | tail -3 ) | tee ${L}_pytest.log | cut -c1-200
for v in 1 0 1 0; do STX_WGRAD_BAL=$v timeout 200 python tools/kernel_bench.py --iters 30 --only wgrad 2>&1 | grep '"kernel"' | grep -v "c1\|1x1" | sed "s/^/bal=$v /" | tee -a ${L}_wgrad.txt | cut -c1-120; done
for v in 1 0; do STX_WGRAD_BAL=$v timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | cut -c1-200 | sed "s/^/bal=$v /" | tee -a ${L}_bench.txt; done
