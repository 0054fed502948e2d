#!/bin/bash
# Round 4, call D: the whole GPU suite on the current build; kernel table cold + warm with the round's A/B sets; bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4d
R=$PWD
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 ) > ${L}_pytest.log 2>&1; tail -8 ${L}_pytest.log | cut -c1-400
timeout 400 python tools/kernel_bench.py --iters 20 --cold --ab --ab-filter "march" > ${L}_kernel_bench_cold.log 2>&1; grep -E '"kernel"|"ab"' ${L}_kernel_bench_cold.log > ${L}_kernel_bench_cold.jsonl; cut -c1-150 ${L}_kernel_bench_cold.jsonl
timeout 300 python tools/kernel_bench.py --iters 20 > ${L}_kernel_bench_warm.log 2>&1; grep -E '"kernel"' ${L}_kernel_bench_warm.log > ${L}_kernel_bench_warm.jsonl; wc -l ${L}_kernel_bench_warm.jsonl
for l1 in 0 1; do STX_CONV_L1_MARCH=$l1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_l1march$l1.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_l1march$l1.json"))
print("L1_MARCH=$l1", d["ms_per_step"], d["roofline"]["frac"], d["roofline_volume_build"]["avg_launch_ms"], d["roofline_volume_build"]["frac"], d.get("hot_path_ms"), d.get("feature_cnn_ms"))
EOF2
done
