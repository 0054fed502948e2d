#!/bin/bash
# Round 2, GPU call P: head backward generations apart (per-pixel pass / gather), cost-volume backward schedules x
# prefetch depth, sampled volume with the left features in registers; true durations of the small kernels from a trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
( timeout 120 python -m pytest tests -m gpu -q -p no:cacheprovider -k "head or sampled or (cost_volume_bwd)" > gpurun_out/pytest_gpu_p.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_p.log )
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/pytest_gpu_p.log | tail -8
echo "[t=$(el)s] pytest done"
timeout 120 python tools/kernel_bench.py --iters 20 --ab --only head,cost_volume,sampled_volume,bn_finalize > gpurun_out/kernel_bench_p.log 2>&1; grep -E '"kernel"|"ab"' gpurun_out/kernel_bench_p.log > gpurun_out/kernel_bench_p.jsonl; cut -c1-125 gpurun_out/kernel_bench_p.jsonl
echo "[t=$(el)s] A/B done"
for v in "X=0" "STX_HEAD_V1=7 STX_BN_FINALIZE_V1=1 STX_WGRAD_REDUCE_V1=1"; do
  tag=$(echo $v | cut -c1-10 | tr '= ' '__')
  ( cd /tmp && env $v timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/tr_$tag -o tr --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 5 --only head,bn_finalize,conv_32_32_L0_wgrad > /dev/null 2>&1 )
  echo "== [$v]" >> gpurun_out/trace_small_p.txt
  python tools/rocprof_summary.py /tmp/tr_$tag 2>&1 | grep -E "head|bn_finalize|wgrad" >> gpurun_out/trace_small_p.txt
done
cut -c1-170 gpurun_out/trace_small_p.txt
echo "[t=$(el)s] end"
