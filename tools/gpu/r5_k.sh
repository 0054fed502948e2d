#!/bin/bash
# Round 5, call K: transposed 128->64 with wave-uniform epilogue (A/B against the previous build), kernel tests, headline.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5k
B=$PWD/stereo_toolbox_amd/lib/libstx_hip_before.so
for rep in 1 2; do
  STX_BENCH_LIB=$B timeout 200 python tools/kernel_bench.py --cold --iters 20 --only deconv 2>/dev/null | sed "s/^/before$rep /" | cut -c1-150
  timeout 200 python tools/kernel_bench.py --cold --iters 20 --only deconv 2>/dev/null | sed "s/^/after$rep  /" | cut -c1-150
done > ${L}_deconv_ab.txt 2>&1; sort -k3,3 -s ${L}_deconv_ab.txt | cut -c1-150
( timeout 300 python -m pytest tests/test_kernels.py -x -q -m gpu -p no:cacheprovider -k "deconv or dgrad" 2>&1 | tail -2 ) 2>&1 | cut -c1-200
timeout 300 python bench.py --config gwc_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_gwc.json"))
print("gwc_train", d["value"], d["ms_per_step"], d.get("hot_path_ms"), d.get("feature_cnn_ms"))
EOF2
