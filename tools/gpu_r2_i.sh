#!/bin/bash
# Round 2, GPU call I: matrix-core cost-volume backward -- parity, A/B against the first generation, PMC.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_models.py -m gpu -q -p no:cacheprovider -k "cost_volume or gwcnet_gc_train or cfnet_train or acvnet_train" > gpurun_out/pytest_i.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_i.log | tail -8
for v in "STX_CVB_OLD=1" "" "STX_CVB_GRID=128" "STX_CVB_GRID=512"; do
  echo "== cost volume bwd variant [$v]" | tee -a gpurun_out/cvb_ab.log
  env $v STX_CVB_TRACE=1 timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep -E "kernel|stx\]" | sort | uniq -c | tee -a gpurun_out/cvb_ab.log | cut -c1-150
done
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmci_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only cost_volume > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmci_$tag cost_volume_bwd > /root/repo/gpurun_out/pmc_cvb_$tag.txt 2>&1
done
cat /root/repo/gpurun_out/pmc_cvb_*.txt | cut -c1-150
cd /root/repo
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_i.log | cut -c1-330
