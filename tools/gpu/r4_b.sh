#!/bin/bash
# Round 4, call B: IGEV aggregation + DDP / RCCL world-1 tests on the GPU, fetch counters of the cost-volume forward per
# prefetch scheme (call A asked kernel_bench for a name its section filter did not match).
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4b
R=$PWD
( timeout 900 python -m pytest tests/test_igev_aggregation.py tests/test_trainer_dropin.py tests/test_distributed.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40 ) > ${L}_pytest.log 2>&1; tail -12 ${L}_pytest.log | cut -c1-400
for pf in 2 1; do for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && STX_CV_PF=$pf timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 5 --only cost_volume > /dev/null 2>&1 )
  echo "== STX_CV_PF=$pf" >> ${L}_pmc_cv_fwd.txt; python tools/pmc_summary.py /tmp/pmc_x cost_volume >> ${L}_pmc_cv_fwd.txt 2>&1
done; done
cut -c1-120 ${L}_pmc_cv_fwd.txt
