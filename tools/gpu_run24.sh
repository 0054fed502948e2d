#!/bin/bash
export TMPDIR=/tmp
timeout 120 python tools/kernel_bench.py --iters 10 --only c1 2>&1 | grep '"kernel"' | cut -c1-100
timeout 300 python -m pytest tests/test_kernels.py -m gpu -k "c1" -q 2>&1 | tail -2
