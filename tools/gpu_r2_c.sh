#!/bin/bash
# Round 2, GPU call C: cost-volume builder v3 (macro-units + register ring) A/B, PMC of the builder and of the backward,
# bench line with the new defaults.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "cost_volume or conv3d or deconv or dgrad" > gpurun_out/pytest_c.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_c.log | tail -8
for v in "" "STX_CV_NT=0" "STX_CV_NSW=4" "STX_CV_QPW=2" "STX_CV_WGS=1" "STX_CV_WGS=2 STX_CV_NSW=4"; do
  echo "== cost volume variant [$v]" | tee -a gpurun_out/cv_ab3.log
  env $v timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep kernel | tee -a gpurun_out/cv_ab3.log | cut -c1-120
done
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmcv_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only cost_volume > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmcv_$tag cost_volume > /root/repo/gpurun_out/pmc_cv_$tag.txt 2>&1
done
cat /root/repo/gpurun_out/pmc_cv_*.txt | cut -c1-150
cd /root/repo
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_c.log | cut -c1-330
