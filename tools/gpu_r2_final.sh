#!/bin/bash
# Round 2 validation: full GPU test-suite, smoke, headline bench line (+ the other BASELINE configs), kernel table.
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_final.log )
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/pytest_gpu_final.log | tail -12
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_final.log 2>&1; tail -2 gpurun_out/smoke_final.log
timeout 900 python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-2600
for c in psm_volume kitti_infer acv_train; do
  timeout 900 python bench.py --config $c --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/bench_configs_final.log | cut -c1-700
done
timeout 900 python tools/kernel_bench.py --iters 10 > gpurun_out/kernel_bench_final.log 2>&1; grep kernel gpurun_out/kernel_bench_final.log > gpurun_out/kernel_bench_final.jsonl; cat gpurun_out/kernel_bench_final.jsonl | cut -c1-110
