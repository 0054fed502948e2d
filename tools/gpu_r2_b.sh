#!/bin/bash
# Round 2, GPU call B: cost-volume builder v2 (role-specialised waves) A/B, pipelined conv kernel A/B, march W4 A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "cost_volume or conv3d or deconv or dgrad" > gpurun_out/pytest_b.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_b.log | tail -8
for v in "" "STX_CV_NT=1" "STX_CV_QPW=2" "STX_CV_NSW=4" "STX_CV_NSW=12" "STX_CV_WGS=1" "STX_CV_NSW=12 STX_CV_NT=1"; do
  echo "== cost volume variant [$v]" | tee -a gpurun_out/cv_ab2.log
  env $v timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep kernel | tee -a gpurun_out/cv_ab2.log | cut -c1-120
done
for v in "STX_CONV_PIPE=0" "STX_CONV_PIPE=1" "STX_CONV_PIPE=1 STX_CONV_PIPE_WGS=1" "STX_MARCH_W4=1"; do
  echo "== conv variant [$v]" | tee -a gpurun_out/conv_ab.log
  env $v timeout 300 python tools/kernel_bench.py --iters 10 --only _fwd --skip-wgrad 2>&1 | grep -E "conv_|deconv" | tee -a gpurun_out/conv_ab.log | cut -c1-120
done
for v in "STX_CONV_PIPE=1" "STX_CONV_PIPE=0" "STX_MARCH_W4=1"; do
  echo "== bench [$v]" | tee -a gpurun_out/bench_b.log
  env $v timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/bench_b.log | cut -c1-330
done
