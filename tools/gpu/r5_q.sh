#!/bin/bash
# Round 5, call Q: the march forward kernel with a wave-uniform wave index, A/B against the committed build (alternating processes)
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5q
B=$PWD/stereo_toolbox_amd/lib/libstx_hip_before.so
K=conv_32_32_L0_fwd,conv_64_32_L0_fwd
for rep in 1 2 3; do
  STX_BENCH_LIB=$B timeout 200 python tools/kernel_bench.py --cold --iters 30 --only $K 2>/dev/null | sed "s/^/before$rep /" | cut -c1-150
  timeout 200 python tools/kernel_bench.py --cold --iters 30 --only $K 2>/dev/null | sed "s/^/after$rep  /" | cut -c1-150
done > ${L}_march_uniform_ab.txt 2>&1; sort -k3,3 -s ${L}_march_uniform_ab.txt | cut -c1-150
