#!/bin/bash
# Round 2, GPU call Q: transposed-conv kernel with the straight-line, weight-prefetching tap list vs the rolled loops.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 120 python -m pytest tests -m gpu -q -p no:cacheprovider -k "head or deconv or dgrad or layer_shapes" > gpurun_out/pytest_gpu_q.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_q.log )
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/pytest_gpu_q.log | tail -8
timeout 120 python tools/kernel_bench.py --iters 20 --ab --only deconv,head > gpurun_out/kernel_bench_q.log 2>&1; grep -E '"kernel"|"ab"' gpurun_out/kernel_bench_q.log | grep -E "deconv|head|ab" > gpurun_out/kernel_bench_q.jsonl; grep -B1 -E "deconv" gpurun_out/kernel_bench_q.jsonl | cut -c1-125; grep -A2 '"head' gpurun_out/kernel_bench_q.jsonl | head -8 | cut -c1-125
