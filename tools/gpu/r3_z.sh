#!/bin/bash
# Round 3, GPU call Z: bench lines of the final defaults (fused Adam) for profiles/r03_bench_line.json / r03_bench_configs.jsonl.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3z
for c in gwc_train acv_train kitti_infer psm_volume; do timeout 600 python bench.py --config $c $( [ $c = gwc_train ] || echo --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; cut -c1-200 ${L}_bench_$c.json; done
timeout 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --graph 2>&1 | grep '^{' | tail -1 | cut -c1-200 | sed "s/^/graph /" | tee ${L}_bench_graph.txt
