#!/bin/bash
# Round 4, call J: implicit-GEMM tap loops with the k-step outermost (consecutive MFMAs on different accumulators): parity + cold table.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4j
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "conv3d or deconv" 2>&1 | grep -v "^  " | tail -20 ) > ${L}_pytest.log 2>&1; grep -E "passed|failed|^FAILED|^E  " ${L}_pytest.log | cut -c1-300
timeout 300 python tools/kernel_bench.py --iters 20 --cold --only conv_,deconv_ --skip-wgrad > ${L}_kb_cold.log 2>&1; grep -E '"kernel"|"ab"' ${L}_kb_cold.log | cut -c1-200
