#!/bin/bash
export TMPDIR=/tmp
run() { timeout 120 python tools/kernel_bench.py --iters 5 --only "$1" 2>&1 | grep '"kernel"' | cut -c1-100; }
for ab in 0 1 2 4 8 3 15 14; do echo "=== ablate $ab"; STX_CVB_ABLATE=$ab run cost_volume | grep bwd; done
