#!/bin/bash
# Round 3, GPU call U: straight-line epilogue of the implicit-GEMM / pipelined / transposed kernels.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3u
( timeout 900 python -m pytest tests/test_kernels.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) | tee ${L}_pytest.log | cut -c1-200
timeout 300 python tools/kernel_bench.py --iters 20 --only conv_,deconv,conv1x1 2>&1 | grep '"kernel"' | grep -v wgrad | tee ${L}_kernel_bench.jsonl | cut -c1-110
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench.json | cut -c1-300
timeout 500 python bench.py --config kitti_infer --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench_kitti.json | cut -c1-300
