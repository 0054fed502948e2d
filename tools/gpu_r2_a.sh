#!/bin/bash
# Round 2, GPU call A: full GPU test-suite (new full-size parity tests, PCWNet), stream ceiling, cost-volume A/B, bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log ) 
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/pytest_gpu.log | tail -15
timeout 120 tools/ubench/store_stream > gpurun_out/store_stream.log 2>&1; cat gpurun_out/store_stream.log
for v in "" "STX_CV_NT=1" "STX_CV_QPW=2" "STX_CV_QPW=2 STX_CV_NT=1" "STX_CV_WGS=1" "STX_CV_WGS=3" "STX_CV_OLD=1"; do
  echo "== cost volume variant [$v]" | tee -a gpurun_out/cv_ab.log
  env $v timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep kernel | tee -a gpurun_out/cv_ab.log | cut -c1-120
done
timeout 900 python bench.py > gpurun_out/bench_a.log 2>&1; tail -1 gpurun_out/bench_a.log | cut -c1-2500
