#!/bin/bash
# Round 3, GPU call W: weight gradient in the 128-VGPR form, two workgroups per CU (STX_WGRAD_OCC2).
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3w
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -3 ) | tee ${L}_pytest.log | cut -c1-200
for v in 0 1 2 0 1 2; do STX_WGRAD_OCC2=$v timeout 120 python tools/kernel_bench.py --iters 30 --only conv_32_32_L0_wgrad,conv_64_32_L0_wgrad 2>&1 | grep '"kernel"' | sed "s/^/occ2=$v /" | tee -a ${L}_wgrad.txt; done
