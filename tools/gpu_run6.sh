#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== cv"; timeout 600 python tools/kernel_bench.py --iters 5 --only cost_volume 2>&1 | grep -v amdgpu.ids
echo "== CK16"; STX_CONV_CK=16 timeout 600 python tools/kernel_bench.py --iters 5 --only _fwd 2>&1 | grep -E "conv_(64_32|32_32|64_64|128_128)"
echo "== CK8"; STX_CONV_CK=8 timeout 600 python tools/kernel_bench.py --iters 5 --only _fwd 2>&1 | grep -E "conv_(64_32|32_32|64_64)"
echo "== WGRAD 8 waves"; STX_WGRAD_WAVES=8 timeout 600 python tools/kernel_bench.py --iters 5 --only wgrad 2>&1 | grep wgrad
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/pytest_gpu.log | tail -5
