#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r20.log 2>&1; tail -1 gpurun_out/bench_r20.log | cut -c1-700
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/rocprof_bench.log 2>&1
cd /root/repo; python tools/rocprof_summary.py /tmp/prof_bench > gpurun_out/prof_bench_summary_r20.txt 2>&1; head -45 gpurun_out/prof_bench_summary_r20.txt | cut -c1-160
