#!/bin/bash
# Round 3, GPU call S: weight gradient with two tiles in LDS (STX_WGRAD_DB).
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3s
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "wgrad or block" 2>&1 | tail -3 ) | tee ${L}_pytest.log | cut -c1-200
for v in 1 0 1 0; do STX_WGRAD_DB=$v timeout 120 python tools/kernel_bench.py --iters 30 --only conv_32_32_L0_wgrad,conv_64_32_L0_wgrad 2>&1 | grep '"kernel"' | sed "s/^/db=$v /" | tee -a ${L}_wgrad.txt; done
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench.json | cut -c1-300
