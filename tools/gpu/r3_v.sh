#!/bin/bash
# Round 3, GPU call V: inference step trace (kitti_infer) after the epilogue / glue changes.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3v
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_inf -o bench --output-format csv -- python $R/bench.py --config kitti_infer --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_infer.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_inf --steady cost_volume_fwd 3 > ${L}_infer_kernel_trace_steady.txt 2>&1; head -36 ${L}_infer_kernel_trace_steady.txt | cut -c1-150
