#!/bin/bash
# Round 3, GPU call Y: streaming 1x1x1 convolution (32 channels); torch.optim.Adam(fused=True) in the bench step.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3y
( timeout 900 python -m pytest tests/test_kernels.py tests/test_hygiene.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) | tee ${L}_pytest.log | cut -c1-200
timeout 300 python tools/kernel_bench.py --iters 30 --only conv1x1 2>&1 | grep '"kernel"' | tee ${L}_kernel_bench.jsonl | cut -c1-110
for f in 1 0 1 0; do timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --fused-adam $f 2>&1 | grep '^{' | tail -1 | cut -c1-200 | sed "s/^/fused_adam=$f /" | tee -a ${L}_bench.txt; done
timeout 500 python bench.py --config kitti_infer --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench_kitti.json | cut -c1-250
