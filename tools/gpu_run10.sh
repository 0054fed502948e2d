#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/pytest_gpu.log | tail -6
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python tools/kernel_bench.py --iters 5 --only cost_volume 2>&1 | grep -v amdgpu
