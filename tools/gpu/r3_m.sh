#!/bin/bash
# Round 3, GPU call M: classifier-tail forward with the next plane in flight; weight-gradient loop back to the conditional form; small-shape train parity.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3m
( timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 ) | tee ${L}_pytest.log | cut -c1-200
( timeout 300 python -m pytest tests/test_models.py -m gpu -q -p no:cacheprovider --tb=short -k "gwcnet_gc_train_parity or acvnet_train_parity" 2>&1 | grep -E "grad err|passed|failed|Error|train_grads" | cut -c1-400 ) | tee -a ${L}_small_train.txt
timeout 300 python tools/kernel_bench.py --iters 20 --only wgrad,c1 > ${L}_kernel_bench.log 2>&1; grep -E '"kernel"' ${L}_kernel_bench.log | cut -c1-110
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | tee ${L}_bench.json | cut -c1-400
