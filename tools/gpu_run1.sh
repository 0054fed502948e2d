#!/bin/bash
# First GPU session: kernel parity tests, per-kernel timings, rocprofv3 kernel stats.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/rocminfo.txt 2>&1
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q > gpurun_out/pytest_kernels.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_kernels.log
tail -15 gpurun_out/pytest_kernels.log
timeout 600 python tools/kernel_bench.py --iters 5 > gpurun_out/kernel_bench.log 2>&1
echo "bench exit $?" >> gpurun_out/kernel_bench.log
cat gpurun_out/kernel_bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof1 -o kb -- python /root/repo/tools/kernel_bench.py --iters 3 > /root/repo/gpurun_out/rocprof1.log 2>&1
find gpurun_out/prof1 -name "*stats*" | head; find gpurun_out/prof1 -name "*kernel_stats*" -exec head -40 {} \;
