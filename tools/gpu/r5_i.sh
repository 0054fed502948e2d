#!/bin/bash
# Round 5, call I: wave index made wave-uniform (readfirstlane) in the implicit-GEMM / pipelined / transposed kernels -- no
# waterfall loops around the buffer descriptors, scalar address math: same-box A/B against the previous build (two processes per
# library, alternating), then the kernel tests, the bench lines.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5i
B=$PWD/stereo_toolbox_amd/lib/libstx_hip_before.so
K=conv_64_64_L1_fwd,conv_32_64_s2_L0_fwd,conv_128_128_L2_fwd,conv_64_128_s2_L1_fwd,deconv,conv1x1
for rep in 1 2; do
  STX_BENCH_LIB=$B timeout 200 python tools/kernel_bench.py --cold --iters 20 --only $K 2>/dev/null | sed "s/^/before$rep /" | cut -c1-150
  timeout 200 python tools/kernel_bench.py --cold --iters 20 --only $K 2>/dev/null | sed "s/^/after$rep  /" | cut -c1-150
done > ${L}_uniform_wave_ab.txt 2>&1; sort -k3,3 -s ${L}_uniform_wave_ab.txt | cut -c1-150
( timeout 300 python -m pytest tests/test_kernels.py -x -q -m gpu -p no:cacheprovider -k "conv3d or deconv or dgrad" 2>&1 | tail -2 ) 2>&1 | cut -c1-200
for lib in before after; do for c in gwc_train kitti_infer; do
  if [ $lib = before ]; then export STX_HIP_LIB=$B; else unset STX_HIP_LIB; fi
  timeout 300 python bench.py --config $c --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_${c}_$lib.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_${c}_$lib.json"))
print("$c $lib", d["value"], d["ms_per_step"], d.get("hot_path_ms"), d.get("feature_cnn_ms"))
EOF2
done; done
