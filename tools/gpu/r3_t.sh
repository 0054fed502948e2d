#!/bin/bash
# Round 3, GPU call T: classifier-tail data gradient on the matrix cores; inference with the fused 2-D BatchNorm glue.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3t
( timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "c1 or block" 2>&1 | tail -3 ) | tee ${L}_pytest.log | cut -c1-200
( timeout 900 python -m pytest tests/test_models.py -m gpu -q -p no:cacheprovider -k "eval" 2>&1 | tail -12 ) | tee ${L}_pytest_eval.log | cut -c1-330
timeout 300 python tools/kernel_bench.py --iters 30 --only c1 2>&1 | grep '"kernel"' | tee ${L}_kernel_bench.jsonl | cut -c1-110
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench.json | cut -c1-300
for f in 1 0 1 0; do STX_FEAT2D_FUSED=$f timeout 500 python bench.py --config kitti_infer --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | cut -c1-220 | sed "s/^/fused=$f /" | tee -a ${L}_bench_kitti.txt; done
