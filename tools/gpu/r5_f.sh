#!/bin/bash
# Round 5, call F: tap-loop micro-benchmark with the MT=2 / NT=1 wave mapping (half the weight loads per MFMA)
mkdir -p gpurun_out
timeout 120 tools/ubench/igemm_loop > gpurun_out/r5f_ubench_igemm_loop.txt 2>&1; cat gpurun_out/r5f_ubench_igemm_loop.txt
