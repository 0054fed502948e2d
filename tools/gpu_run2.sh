#!/bin/bash
# GPU session: full gpu test-suite, smoke, bench line, rocprof of bench (raw trace stays in /tmp).
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|Error" gpurun_out/pytest_gpu.log | tail -15
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -4 gpurun_out/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1; echo "bench exit $?" >> gpurun_out/bench1.log; tail -2 gpurun_out/bench1.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/rocprof_bench.log 2>&1
cd /root/repo; find /tmp/prof_bench -type f | head -5; python tools/rocprof_summary.py /tmp/prof_bench > gpurun_out/prof_bench_summary.txt 2>&1; head -45 gpurun_out/prof_bench_summary.txt
