#!/bin/bash
# Round 2, GPU call D: cost-volume builder with the lean store-wave flush, f-4 tests on the GPU, kernel trace of the bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py tests/test_igev_preprocess.py -m gpu -q -p no:cacheprovider -k "cost_volume or igev or pad_normalize or prepare" > gpurun_out/pytest_d.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_d.log | tail -8
for v in "" "STX_CV_NSW=4" "STX_CV_WGS=1" "STX_CV_NT=0"; do
  echo "== cost volume variant [$v]" | tee -a gpurun_out/cv_ab4.log
  env $v timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep kernel | tee -a gpurun_out/cv_ab4.log | cut -c1-120
done
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/rocprof_bench_d.log 2>&1
cd /root/repo; python tools/rocprof_summary.py /tmp/prof_bench > gpurun_out/prof_bench_summary_d.txt 2>&1; head -50 gpurun_out/prof_bench_summary_d.txt | cut -c1-170
tail -1 gpurun_out/rocprof_bench_d.log | cut -c1-300
