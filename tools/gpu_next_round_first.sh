#!/bin/bash
# FIRST GPU call of the next round (≈ 40 s on a warm box): everything that was written after the GPU budget of round 2 ran
# out and is therefore parity-tested on the emulator only.
#   * CFNet cascade-volume backward with the LDS window (default) vs global atomics (STX_SV_BWD_V1=1, 1.62 ms in round 2)
#   * stride-2 32->64 conv with the dense LDS tile (STX_CONV_S2_DENSE=1, opt-in) vs the padded one (0.328 ms)
#   * the complete kernel table and the A/B sets of tools/kernel_bench.py
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 120 python -m pytest tests -m gpu -q -p no:cacheprovider -k "sampled or stride2_dense or head or deconv" 2>&1 | tail -6 ) > gpurun_out/pytest_gpu_next.log 2>&1; cat gpurun_out/pytest_gpu_next.log
timeout 150 python tools/kernel_bench.py --iters 10 --ab > gpurun_out/kernel_bench_next.log 2>&1; grep -E '"kernel"|"ab"' gpurun_out/kernel_bench_next.log > gpurun_out/kernel_bench_next.jsonl; cut -c1-130 gpurun_out/kernel_bench_next.jsonl
