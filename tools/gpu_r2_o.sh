#!/bin/bash
# Round 2, GPU call O (the last GPU minutes of the round): the kernels added after call N on the real chip.
#   1 GPU tests of what changed (head generations, BN finalize, slab reduce, sampled volume, CFNet eval, cost volume) and
#     the full-size GwcNet_GC train-step parity test as the whole-path check of the new defaults
#   2 kernel table + in-process A/B of every per-call switch      3 PMC of the cost-volume kernels      4 cfg2 bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
left() { [ $(el) -lt $1 ]; }
rm -f gpurun_out/parity_report.jsonl
( timeout 170 python -m pytest tests -m gpu -q -p no:cacheprovider -k "head or bn_finalize or wgrad or sampled or cfnet_eval or full_size_train or (cost_volume and not full_size)" > gpurun_out/pytest_gpu_o.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_o.log )
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/pytest_gpu_o.log | tail -12
echo "[t=$(el)s] pytest done"
timeout 150 python tools/kernel_bench.py --iters 10 --ab > gpurun_out/kernel_bench_o.log 2>&1; grep -E '"kernel"|"ab"' gpurun_out/kernel_bench_o.log > gpurun_out/kernel_bench_o.jsonl; cut -c1-125 gpurun_out/kernel_bench_o.jsonl
echo "[t=$(el)s] kernel table + A/B done"
if left 290; then
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | cut -d' ' -f1)
    ( cd /tmp && timeout 60 rocprofv3 --pmc $grp -d /tmp/pmco_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only cost_volume > /dev/null 2>&1 )
    python tools/pmc_summary.py /tmp/pmco_$tag cost_volume >> gpurun_out/pmc_cv_o.txt 2>&1
    left 330 || break
  done
  cut -c1-150 gpurun_out/pmc_cv_o.txt
  echo "[t=$(el)s] pmc done"
fi
if left 340; then
  timeout 60 python bench.py --config psm_volume --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/bench_configs_o.log | cut -c1-700
fi
echo "[t=$(el)s] end"
