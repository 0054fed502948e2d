#!/bin/bash
# Round 6, call F: write-pattern ceilings for the volume builder (tools/ubench/store_pattern) + the round's starting step trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r6f
R=$PWD
./tools/ubench/store_pattern > ${L}_store_pattern.txt 2>&1; tail -3 ${L}_store_pattern.txt
./tools/ubench/store_stream > ${L}_store_stream.txt 2>&1
timeout 700 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc_train.json; cut -c1-170 ${L}_bench_gwc_train.json
( cd /tmp && rm -rf /tmp/prof_bench && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -8 ${L}_bench_kernel_trace_steady.txt | cut -c1-150
