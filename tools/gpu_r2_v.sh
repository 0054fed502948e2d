#!/bin/bash
# Round 2, GPU call V (last GPU seconds): model-level GPU tests with the final defaults, the rest of the kernel tests,
# final kernel table.
mkdir -p gpurun_out
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
rm -f gpurun_out/parity_report.jsonl
timeout 60 python tools/kernel_bench.py --iters 10 > gpurun_out/kernel_bench_v.log 2>&1; grep -E '"kernel"' gpurun_out/kernel_bench_v.log > gpurun_out/kernel_bench_v.jsonl; wc -l gpurun_out/kernel_bench_v.jsonl; grep -E "mish|deconv|conv_64_64|head|bwd_gwc" gpurun_out/kernel_bench_v.jsonl | cut -c1-110
echo "[t=$(el)s] table"
( timeout 35 python -m pytest tests/test_kernels.py tests/test_metrics.py tests/test_igev_preprocess.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4 ) > gpurun_out/pytest_gpu_v_kernels.log 2>&1; cat gpurun_out/pytest_gpu_v_kernels.log
echo "[t=$(el)s] kernels"
( timeout 110 python -m pytest tests/test_models.py -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -22 ) > gpurun_out/pytest_gpu_v_models.log 2>&1; cat gpurun_out/pytest_gpu_v_models.log | cut -c1-200
echo "[t=$(el)s] end"
