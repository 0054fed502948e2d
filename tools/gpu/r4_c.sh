#!/bin/bash
# Round 4, call C: IGEV aggregation + DDP / RCCL world-1 tests on the GPU; march weight-gradient kernel: parity, A/B against
# the tile kernel (warm and cold), counters; bench lines with the cost-volume prefetch schemes.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4c
R=$PWD
( timeout 900 python -m pytest tests/test_igev_aggregation.py tests/test_trainer_dropin.py tests/test_distributed.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40 ) > ${L}_pytest.log 2>&1; tail -12 ${L}_pytest.log | cut -c1-600
( timeout 600 python -m pytest tests/test_kernels.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider -k "wgrad or conv or repro" 2>&1 | tail -12 ) > ${L}_pytest_k.log 2>&1; tail -4 ${L}_pytest_k.log | cut -c1-300
timeout 300 python tools/kernel_bench.py --iters 20 --only wgrad --ab --ab-filter "tile kernel" > ${L}_wgrad_warm.log 2>&1; grep -E '"kernel"|"ab"' ${L}_wgrad_warm.log | cut -c1-170
timeout 300 python tools/kernel_bench.py --iters 20 --cold --only wgrad --ab --ab-filter "tile kernel" > ${L}_wgrad_cold.log 2>&1; grep -E '"kernel"|"ab"' ${L}_wgrad_cold.log | cut -c1-170
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only conv_32_32_L0_wgrad,conv_64_64_L1_wgrad > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x wgrad_march >> ${L}_pmc_wgrad_march.txt 2>&1
done
cut -c1-110 ${L}_pmc_wgrad_march.txt
for pf in 2 1; do STX_CV_PF=$pf timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_pf$pf.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_pf$pf.json"))
print("PF=$pf", d["ms_per_step"], d["roofline"]["frac"], d["roofline_volume_build"]["avg_launch_ms"], d["roofline_volume_build"]["frac"], d.get("hot_path_ms"), d.get("feature_cnn_ms"))
EOF2
done
