#!/bin/bash
# Round 4, call I: stride-2 march weight gradient with the staging loads spread over the MFMA groups: parity + cold A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4i
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "wgrad or deconv" 2>&1 | grep -v "^  " | tail -20 ) > ${L}_pytest_wgrad.log 2>&1; grep -E "passed|failed|^FAILED|^E  " ${L}_pytest_wgrad.log | cut -c1-300
timeout 300 python tools/kernel_bench.py --iters 20 --cold --only s2_L0_wgrad,s2_L1_wgrad --ab --ab-filter "s2 / transposed" > ${L}_kb_cold.log 2>&1; grep -E '"kernel"|"ab"' ${L}_kb_cold.log | cut -c1-200
