#!/bin/bash
# Round 4, call K: resident workgroups per CU of the implicit-GEMM convolution launches (hipOccupancyMaxActiveBlocksPerMultiprocessor)
mkdir -p gpurun_out
export TMPDIR=/tmp
STX_REPORT_OCCUPANCY=1 timeout 300 python tools/kernel_bench.py --iters 2 --only conv_32_64_s2,conv_64_64_L1,conv_64_128_s2,conv_128_128_L2,deconv --skip-wgrad 2>&1 | grep -E '"kernel"|\[stx\]' | sort | uniq -c | cut -c1-200 > gpurun_out/r4k_occupancy.log
cat gpurun_out/r4k_occupancy.log
