#!/bin/bash
# Round 3, GPU call J: straight-line epilogue + descriptor staging of the march kernel.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3j
( timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "conv3d_fwd or march or reproducib" 2>&1 | tail -4 ) | tee ${L}_pytest.log | cut -c1-200
for e in 1 0 1 0; do STX_MARCH_EPI=$e timeout 120 python tools/kernel_bench.py --iters 30 --only conv_32_32_L0_fwd,conv_64_32_L0_fwd 2>&1 | grep '"kernel"' | sed "s/^/epi=$e /" | tee -a ${L}_march.txt; done
for a in 1 2; do STX_MARCH_ABLATE=$a timeout 120 python tools/kernel_bench.py --iters 30 --only conv_32_32_L0_fwd 2>&1 | grep '"kernel"' | sed "s/^/ablate=$a /" | tee -a ${L}_march.txt; done
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench.json | cut -c1-400
