#!/bin/bash
# Round 4, call A: the new GPU tests (reference-trainer drop-in / DDP / RCCL world-1, split_mode, deterministic ac_volume_bwd,
# cost-volume forward schedules), cold-cache A/B of the cost-volume forward, its fetch counters per prefetch scheme, and a
# bench line with the hot-path / feature-CNN split.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4a
R=$PWD
( timeout 600 python -m pytest tests/test_trainer_dropin.py tests/test_distributed.py tests/test_kernels.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 ) > ${L}_pytest.log 2>&1; tail -3 ${L}_pytest.log | cut -c1-300
( timeout 300 python -m pytest tests/test_models.py -m gpu -q -p no:cacheprovider -k "functional or gwcnet_gc_train or eval_parity" 2>&1 | tail -8 ) > ${L}_pytest_models.log 2>&1; tail -2 ${L}_pytest_models.log | cut -c1-300
timeout 300 python tools/kernel_bench.py --iters 30 --cold --only cost_volume --ab --ab-filter "cost volume fwd" > ${L}_cv_cold.log 2>&1; grep -E '"kernel"|"ab"' ${L}_cv_cold.log > ${L}_cv_cold.jsonl; cut -c1-160 ${L}_cv_cold.jsonl
timeout 200 python tools/kernel_bench.py --iters 30 --only cost_volume > ${L}_cv_warm.log 2>&1; grep -E '"kernel"' ${L}_cv_warm.log | cut -c1-160
for pf in 2 1; do for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && STX_CV_PF=$pf timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 5 --only cost_volume_fwd_gwcgc > /dev/null 2>&1 )
  echo "== STX_CV_PF=$pf" >> ${L}_pmc_cv_fwd.txt; python tools/pmc_summary.py /tmp/pmc_x cost_volume_fwd >> ${L}_pmc_cv_fwd.txt 2>&1
done; done
cut -c1-120 ${L}_pmc_cv_fwd.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench.json; cut -c1-1500 ${L}_bench.json
