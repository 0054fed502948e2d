#!/bin/bash
# Round 2, GPU call J: cost-volume backward register-set depth A/B, estimators, 64->64 on the march kernel.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "cost_volume or estimators or conv3d_fwd" > gpurun_out/pytest_j.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_j.log | tail -8
for v in "STX_CVB_NSET=2" "STX_CVB_NSET=3" "STX_CVB_NSET=4"; do
  echo "== cost volume bwd variant [$v]" | tee -a gpurun_out/cvb_ab2.log
  env $v timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume_bwd,cost_volume 2>&1 | grep -E "kernel" | tee -a gpurun_out/cvb_ab2.log | cut -c1-150
done
echo "== estimators" | tee -a gpurun_out/cvb_ab2.log
timeout 300 python tools/kernel_bench.py --iters 10 --only estimator 2>&1 | grep kernel | tee -a gpurun_out/cvb_ab2.log | cut -c1-120
for v in "STX_MARCH_6464=0" "STX_MARCH_6464=1"; do
  echo "== conv 64->64 [$v]" | tee -a gpurun_out/cvb_ab2.log
  env $v timeout 300 python tools/kernel_bench.py --iters 10 --only conv_64_64_L1_fwd 2>&1 | grep kernel | tee -a gpurun_out/cvb_ab2.log | cut -c1-120
done
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmcj_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only cost_volume > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmcj_$tag cost_volume_bwd > /root/repo/gpurun_out/pmc_cvb2_$tag.txt 2>&1
done
cat /root/repo/gpurun_out/pmc_cvb2_*.txt | cut -c1-150
