#!/bin/bash
# Round 3, GPU call B: blocked sums in the march kernel (speed A/B + error attribution again), full-size parity tests incl. the
# new ACVNet B=2 train step, the hygiene tests on the chip.
mkdir -p gpurun_out
export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=$PWD/stereo_toolbox_amd/tuning/miopen
L=gpurun_out/r3b
for bs in 1 0; do STX_MARCH_BS=$bs timeout 90 python tools/kernel_bench.py --iters 20 --only conv_32_32_L0_fwd,conv_64_32_L0_fwd 2>&1 | grep kernel | sed "s/^/march_bs=$bs /" | tee -a ${L}_march_bs.txt; done
for bs in 1 0; do STX_MARCH_BS=$bs timeout 90 python tools/kernel_bench.py --iters 20 --only conv_32_32_L0_fwd,conv_64_32_L0_fwd 2>&1 | grep kernel | sed "s/^/march_bs=$bs /" | tee -a ${L}_march_bs.txt; done
timeout 300 python tools/parity_isolation.py --tag gwc_gc_384x1248 --label blocked_sums 2>&1 | tail -1 | cut -c1-900
timeout 300 python tools/parity_isolation.py --tag gwc_gc_576x960 --label blocked_sums 2>&1 | tail -1 | cut -c1-900
( timeout 900 python -m pytest tests/test_models.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider -x -k "full_size or reproducibility or train_parity" 2>&1 | tail -25 ) > ${L}_pytest.log 2>&1; cat ${L}_pytest.log | cut -c1-1200
