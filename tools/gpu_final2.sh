#!/bin/bash
# Re-validation after the last kernel changes: gpu tests, smoke, bench line, kernel table.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_gpu.log | tail -8
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-1800
timeout 600 python tools/kernel_bench.py --iters 5 > gpurun_out/kernel_bench.log 2>&1; grep kernel gpurun_out/kernel_bench.log > gpurun_out/kernel_bench.jsonl; grep -E "cost_volume|c1" gpurun_out/kernel_bench.jsonl | cut -c1-110
