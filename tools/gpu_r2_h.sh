#!/bin/bash
# Round 2, GPU call H: full GPU suite after the deconv-wgrad padding fix, LDS-staged modal estimators A/B,
# PMC of the stride-2 / transposed conv kernels (the ones still under 0.60 of the fp32 MFMA peak).
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_h.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_h.log )
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/pytest_gpu_h.log | tail -12
for v in "STX_MODAL_INPLACE=1" ""; do
  echo "== estimators [$v]" | tee -a gpurun_out/estimators_ab.log
  env $v timeout 300 python tools/kernel_bench.py --iters 10 --only estimator 2>&1 | grep kernel | tee -a gpurun_out/estimators_ab.log | cut -c1-120
done
cd /tmp
SEL="conv_32_64_s2,conv_64_128_s2,deconv,conv_64_64_L1,conv_128_128"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmch_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only $SEL > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmch_$tag conv > /root/repo/gpurun_out/pmc_convs2_$tag.txt 2>&1
done
cat /root/repo/gpurun_out/pmc_convs2_*.txt | cut -c1-150 | head -150
cd /root/repo
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_h.log | cut -c1-330
