#!/bin/bash
# Round 4, call H: stride-2 weight gradient on the parity-split march kernel: GPU parity, cold A/B against the tile kernel;
# TORCH_LIBRARY loader tests; the step trace with the bn_* launches split by grid size.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4h
R=$PWD
( timeout 600 python -m pytest tests/test_kernels.py tests/test_torch_ext.py -m gpu -q -p no:cacheprovider -k "wgrad or deconv or torch_ext or ops_match or dispatcher" 2>&1 | grep -v "^  " | tail -20 ) > ${L}_pytest_wgrad.log 2>&1; grep -E "passed|failed|^FAILED|^E  " ${L}_pytest_wgrad.log | cut -c1-300
timeout 300 python tools/kernel_bench.py --iters 20 --cold --only s2_L0_wgrad,s2_L1_wgrad --ab --ab-filter "s2 / transposed" > ${L}_kb_cold.log 2>&1; grep -E '"kernel"|"ab"' ${L}_kb_cold.log | cut -c1-200
( cd /tmp && rm -rf /tmp/tr && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr -o tr --output-format csv -- python $R/bench.py --config gwc_train --steps 6 --warmup 3 --no-cpu-baseline > /tmp/tr.log 2>&1 )
python tools/rocprof_summary.py /tmp/tr --steady cost_volume_fwd 3 --by-grid bn_ > ${L}_trace_bn_by_grid.txt 2>&1
sed -n 1,12p ${L}_trace_bn_by_grid.txt | cut -c1-150; grep -A80 "by launch grid" ${L}_trace_bn_by_grid.txt | cut -c1-140
for c in gwc_train; do timeout 400 python bench.py --config $c --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_$c.json"))
r=d["roofline"]; print("$c", d["value"], d["ms_per_step"], r["frac"], r.get("avg_launch_ms"))
EOF2
done
