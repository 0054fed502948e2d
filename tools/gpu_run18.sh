#!/bin/bash
export TMPDIR=/tmp
run() { timeout 120 python tools/kernel_bench.py --iters 5 --only "$1" 2>&1 | grep '"kernel"' | cut -c1-100; }
echo "=== nopipe 4x16 c512"; run L0_wgrad
echo "=== nopipe 4x16 c256"; STX_WGRAD_CHUNKS=256 run conv_32_32_L0_wgrad
echo "=== pipe 4x16 c256"; STX_WGRAD_PIPE=1 run conv_32_32_L0_wgrad
echo "=== pipe 4x16 c512"; STX_WGRAD_PIPE=1 STX_WGRAD_CHUNKS=512 run conv_32_32_L0_wgrad
echo "=== nopipe 2x32 c512"; STX_WGRAD_TW32=1 run conv_32_32_L0_wgrad
echo "=== nopipe 2x32 c256"; STX_WGRAD_TW32=1 STX_WGRAD_CHUNKS=256 run conv_32_32_L0_wgrad
echo "=== all wgrad default"; run wgrad
