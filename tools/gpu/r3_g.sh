#!/bin/bash
# Round 3, GPU call G: interleaved operand loads in the implicit-GEMM / persistent / transposed conv kernels, 8 compute waves in
# the cost-volume backward, fewer partial rows in the small BN passes; whole-step effect.
mkdir -p gpurun_out
export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=$PWD/stereo_toolbox_amd/tuning/miopen
L=gpurun_out/r3g
( timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "interleaved or eight_compute or blocked_sums or bn_stats or cost_volume_fwd_bwd" 2>&1 | tail -6 ) > ${L}_pytest.log 2>&1; cut -c1-300 ${L}_pytest.log
K=conv_64_64_L1_fwd,conv_32_64_s2_L0_fwd,conv_64_128_s2_L1_fwd,conv_128_128_L2_fwd
for v in 0 1 0 1; do STX_CONV_ILV=$v timeout 120 python tools/kernel_bench.py --iters 20 --only $K 2>&1 | grep kernel | sed "s/^/conv_ilv=$v /" | tee -a ${L}_conv_ilv.txt; done
for v in 1 3 1 3; do STX_DECONV_PIPE=$v timeout 120 python tools/kernel_bench.py --iters 20 --only deconv 2>&1 | grep kernel | sed "s/^/deconv_pipe=$v /" | tee -a ${L}_deconv.txt; done
for v in 4 8 4 8; do STX_CVB_NCW=$v timeout 120 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep kernel | grep bwd | sed "s/^/cvb_ncw=$v /" | tee -a ${L}_cvb.txt; done
STX_CVB_NCW=8 STX_CVB_NSET=2 timeout 120 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep kernel | grep bwd | sed "s/^/cvb_ncw=8 nset=2 /" | tee -a ${L}_cvb.txt
STX_CVB_NCW=8 STX_CVB_TEAM=1 timeout 120 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep kernel | grep bwd | sed "s/^/cvb_ncw=8 team=1 /" | tee -a ${L}_cvb.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v Warning | tail -1 | cut -c1-330 | sed "s/^/default /" | tee -a ${L}_bench.txt
STX_CONV_ILV=1 STX_DECONV_PIPE=3 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v Warning | tail -1 | cut -c1-330 | sed "s/^/ilv /" | tee -a ${L}_bench.txt
