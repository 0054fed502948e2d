#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench7.log 2>&1; tail -1 gpurun_out/bench7.log | cut -c1-1200
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/rocprof_bench.log 2>&1
cd /root/repo; python tools/rocprof_summary.py /tmp/prof_bench > gpurun_out/prof_bench_summary.txt 2>&1; head -64 gpurun_out/prof_bench_summary.txt | cut -c1-170
