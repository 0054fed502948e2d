#!/bin/bash
# Round 5, call O: the split-bf16 sized experiment (tools/ubench/bf16_split_mfma: numerics against fp64 + loop rate) and smoke()
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 tools/ubench/bf16_split_mfma > gpurun_out/r5o_bf16_split_mfma.txt 2>&1; cat gpurun_out/r5o_bf16_split_mfma.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
