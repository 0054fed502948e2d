#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== default (march)"; timeout 600 python tools/kernel_bench.py --iters 5 --only conv_ 2>&1 | grep -E "conv_(64_32|32_32|32_64|64_64)"
echo "== no march"; STX_NO_MARCH=1 timeout 600 python tools/kernel_bench.py --iters 5 --only _fwd 2>&1 | grep -E "conv_(32_32)"
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/pytest_gpu.log | tail -5
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench5.log 2>&1; tail -1 gpurun_out/bench5.log | cut -c1-900
