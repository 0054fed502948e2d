#!/bin/bash
# Round 3: raw whole-block epilogue fast path (implicit-GEMM / pipelined / transposed kernels).
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3raw
( timeout 600 python -m pytest tests/test_kernels.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2 ) | cut -c1-200
timeout 300 python tools/kernel_bench.py --iters 30 --only conv_,deconv,conv1x1 2>&1 | grep '"kernel"' | grep -v wgrad | tee ${L}_kernel_bench.jsonl | cut -c1-110
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench.json | cut -c1-200
