#!/bin/bash
# Round 3, GPU call N: next K chunk in flight in the implicit-GEMM / transposed kernels, exact statistics rows (no memset), trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3n
R=$PWD
( timeout 900 python -m pytest tests/test_kernels.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 ) | tee ${L}_pytest.log | cut -c1-200
timeout 300 python tools/kernel_bench.py --iters 20 --only conv_,deconv,bn_ > ${L}_kernel_bench.log 2>&1; grep -E '"kernel"' ${L}_kernel_bench.log | cut -c1-110
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench.json | cut -c1-300
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -45 ${L}_bench_kernel_trace_steady.txt | cut -c1-150
