#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== default"; timeout 600 python tools/kernel_bench.py --iters 5 > gpurun_out/kb_default.log 2>&1; grep -v amdgpu.ids gpurun_out/kb_default.log
echo "== CK16"; STX_CONV_CK=16 timeout 600 python tools/kernel_bench.py --iters 5 --only _fwd > gpurun_out/kb_ck16.log 2>&1; grep -E "conv_(64_32|32_32|64_64|128_128)" gpurun_out/kb_ck16.log
echo "== CK8"; STX_CONV_CK=8 timeout 600 python tools/kernel_bench.py --iters 5 --only _L0_fwd > gpurun_out/kb_ck8.log 2>&1; grep -E "conv_(64_32|32_32)" gpurun_out/kb_ck8.log
echo "== WGRAD 8 waves"; STX_WGRAD_WAVES=8 timeout 600 python tools/kernel_bench.py --iters 5 --only wgrad > gpurun_out/kb_wg8.log 2>&1; grep wgrad gpurun_out/kb_wg8.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/pytest_gpu.log | tail -5
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench4.log 2>&1; tail -1 gpurun_out/bench4.log | cut -c1-700
