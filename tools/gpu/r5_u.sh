#!/bin/bash
# Round 5, call U: rolling-window patch convolution (ACVNet) -- kernel tests, cold A/B in separate processes, the cfg4 step A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5u
( timeout 300 python -m pytest tests/test_kernels.py -x -q -m gpu -p no:cacheprovider -k "dwconv" 2>&1 | tail -2 ) | cut -c1-200
for rep in 1 2; do for r in 0 1; do
  STX_DWCONV_ROLL=$r timeout 200 python tools/kernel_bench.py --cold --iters 20 --only dwconv 2>/dev/null | sed "s/^/roll=$r rep$rep /" | cut -c1-160
done; done > ${L}_dwconv_ab.txt 2>&1; cat ${L}_dwconv_ab.txt
for r in 0 1; do STX_DWCONV_ROLL=$r timeout 400 python bench.py --config acv_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_acv_roll$r.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_acv_roll$r.json"))
print("acv_train STX_DWCONV_ROLL=$r", d["value"], d["ms_per_step"])
EOF2
done
