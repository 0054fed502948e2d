#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/feat2d_bench.py > gpurun_out/feat2d.log 2>&1; grep -v Warning gpurun_out/feat2d.log | tail -6
PYTORCH_MIOPEN_SUGGEST_NHWC=1 timeout 600 python tools/feat2d_bench.py > gpurun_out/feat2d_nhwc_env.log 2>&1; grep -v Warning gpurun_out/feat2d_nhwc_env.log | tail -6
timeout 600 python tools/kernel_bench.py --iters 5 --only cost_volume > gpurun_out/kb_cv.log 2>&1; cat gpurun_out/kb_cv.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench3.log 2>&1; tail -1 gpurun_out/bench3.log | cut -c1-400
