#!/bin/bash
# Round 5, call G: the implicit-GEMM kernels with the new wave grid (STX_CONV_WN = 2 default; 1 = rounds 1-4; 4 for 128 channels),
# cold and warm, same process; kernel tests of the variants; the headline step with STX_CONV_WN = 1 / 2 / 4.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5g
( timeout 300 python -m pytest tests/test_kernels.py -x -q -m gpu -p no:cacheprovider -k "wave_grid or conv3d_fwd or dgrad" 2>&1 | tail -3 ) 2>&1 | cut -c1-200
timeout 300 python tools/kernel_bench.py --cold --iters 10 --only conv_64_64_L1_fwd,conv_32_64_s2_L0_fwd,conv_128_128_L2_fwd,conv_64_128_s2_L1_fwd --ab --ab-filter "implicit GEMM" > ${L}_kernel_bench_cold_conv_wn.jsonl 2>&1; cut -c1-180 ${L}_kernel_bench_cold_conv_wn.jsonl
for wn in 1 2 4; do STX_CONV_WN=$wn timeout 300 python bench.py --config gwc_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc_wn$wn.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_gwc_wn$wn.json"))
print("gwc_train STX_CONV_WN=$wn", d["value"], d["ms_per_step"], d.get("hot_path_ms"), d.get("feature_cnn_ms"))
EOF2
done
for wn in 1 2; do STX_CONV_WN=$wn timeout 300 python bench.py --config kitti_infer --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_kitti_wn$wn.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_kitti_wn$wn.json"))
print("kitti_infer STX_CONV_WN=$wn", d["value"], d["ms_per_step"])
EOF2
done
