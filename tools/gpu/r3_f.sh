#!/bin/bash
# Round 3, GPU call F: per-plane partial accumulators in the march kernel (correct on the chip? speed? parity gain?), read
# interleave of the march kernel, straight-line weight-gradient loop.
mkdir -p gpurun_out
export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=$PWD/stereo_toolbox_amd/tuning/miopen
L=gpurun_out/r3f
timeout 120 python tools/diag_march_bs.py 2>&1 | tail -12 | cut -c1-200 | tee ${L}_diag.txt
( timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "blocked_sums or straight_line or conv3d_fwd or conv3d_wgrad or dgrad" 2>&1 | tail -6 ) > ${L}_pytest.log 2>&1; cut -c1-300 ${L}_pytest.log
for v in "0 0" "1 0" "0 1" "1 1" "0 0" "1 0"; do set -- $v; STX_MARCH_BS=$1 STX_MARCH_ILV=$2 timeout 90 python tools/kernel_bench.py --iters 20 --only conv_32_32_L0_fwd,conv_64_32_L0_fwd 2>&1 | grep kernel | sed "s/^/bs=$1 ilv=$2 /" | tee -a ${L}_march.txt; done
for v in 0 1 0 1; do STX_WGRAD_V2=$v timeout 120 python tools/kernel_bench.py --iters 20 --only _wgrad 2>&1 | grep kernel | grep -v c1 | sed "s/^/wgrad_v2=$v /" | tee -a ${L}_wgrad.txt; done
timeout 300 python tools/parity_isolation.py --tag gwc_gc_384x1248 --label per_plane_partials 2>&1 | tail -1 | cut -c1-700
timeout 300 python tools/parity_isolation.py --tag gwc_gc_576x960 --label per_plane_partials 2>&1 | tail -1 | cut -c1-700
