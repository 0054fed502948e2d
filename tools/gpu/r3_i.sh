#!/bin/bash
# Round 3, GPU call I: one batched pass of the 2-D CNN over both views with per-view BatchNorm statistics (groups) vs two passes.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3i
( timeout 600 python -m pytest tests/test_kernels.py tests/test_models.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider -k "bn_ or train_parity or full_size_train or reproducib" 2>&1 | tail -8 ) > ${L}_pytest.log 2>&1; cut -c1-250 ${L}_pytest.log
for p in 1 0 1 0; do STX_FEAT2D_PAIRED=$p timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v Warning | tail -1 | cut -c1-200 | sed "s/^/paired=$p /" | tee -a ${L}_bench.txt; done
STX_FEAT2D_PAIRED=1 timeout 500 python bench.py --config acv_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v Warning | tail -1 | cut -c1-200 | sed "s/^/acv paired=1 /" | tee -a ${L}_bench.txt
mkdir -p gpurun_out/miopen_db; cp stereo_toolbox_amd/tuning/miopen/*.txt gpurun_out/miopen_db/
