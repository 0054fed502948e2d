#!/bin/bash
# Round 3, GPU call K: descriptor staging in the weight-gradient, implicit-GEMM, pipelined and transposed kernels.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3k
( timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 ) | tee ${L}_pytest.log | cut -c1-200
timeout 300 python tools/kernel_bench.py --iters 20 > ${L}_kernel_bench.log 2>&1; grep -E '"kernel"' ${L}_kernel_bench.log > ${L}_kernel_bench.jsonl; grep -E "conv|deconv" ${L}_kernel_bench.jsonl | cut -c1-110
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench.json | cut -c1-400
