#!/bin/bash
# Round 3, GPU call C: (1) which blocked-sum variant of the march kernel is right on the chip (variant 1 was wrong in call B
# although the emulator passes) and what each costs; (2) fused channels-last BN glue of the 2-D CNN: parity + speed;
# (3) the full-size parity tests and the reproducibility test.
mkdir -p gpurun_out/miopen_db
export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=$PWD/stereo_toolbox_amd/tuning/miopen
L=gpurun_out/r3c
( timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "blocked_sums or bn_stats or conv3d_fwd" 2>&1 | tail -12 ) > ${L}_pytest_bs.log 2>&1; cut -c1-300 ${L}_pytest_bs.log
for bs in 0 1 2 3 4 0; do STX_MARCH_BS=$bs timeout 90 python tools/kernel_bench.py --iters 20 --only conv_32_32_L0_fwd 2>&1 | grep kernel | sed "s/^/march_bs=$bs /" | tee -a ${L}_march_bs.txt; done
for f in 1 0; do STX_FEAT2D_FUSED=$f timeout 400 python tools/feat2d_bench.py --fmt nchw --no-eval 2>&1 | grep "fwd+bwd" | tee -a ${L}_feat2d.txt; done
( timeout 900 python -m pytest tests/test_models.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider -k "train_parity or eval_parity or full_size or reproducibility or psmnet or eval_mode" 2>&1 | tail -30 ) > ${L}_pytest.log 2>&1; cut -c1-1500 ${L}_pytest.log
for f in 1 0; do STX_FEAT2D_FUSED=$f timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v Warning | tail -1 | cut -c1-400 | sed "s/^/fused=$f /" | tee -a ${L}_bench.txt; done
cp stereo_toolbox_amd/tuning/miopen/*.txt gpurun_out/miopen_db/
