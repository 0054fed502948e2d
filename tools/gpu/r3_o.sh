#!/bin/bash
# Round 3, GPU call O: straight-line epilogues with partial sums / residual; inference step trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3o
R=$PWD
( timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 ) | tee ${L}_pytest.log | cut -c1-200
timeout 300 python tools/kernel_bench.py --iters 20 --only conv_64_32_L0_fwd,conv_32_32_L0_fwd 2>&1 | grep -E '"kernel"' | tee ${L}_kernel_bench.jsonl | cut -c1-110
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench.json | cut -c1-300
timeout 500 python bench.py --config kitti_infer --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench_kitti.json | cut -c1-300
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o bench --output-format csv -- python $R/bench.py --config kitti_infer --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_infer.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_inf --steady cost_volume_fwd 3 > ${L}_infer_kernel_trace_steady.txt 2>&1; head -40 ${L}_infer_kernel_trace_steady.txt | cut -c1-150
