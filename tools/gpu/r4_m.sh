#!/bin/bash
# Round 4, call M: evidence on the build with the stride-2 march weight gradient -- whole GPU suite, the four BASELINE bench lines
# (headline with the CPU baseline), GwcNet_GC step trace (+ the same with the cost-volume backward's team schedule), cold kernel table.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4m
R=$PWD
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^  " | tail -60 ) > ${L}_pytest.log 2>&1; tail -4 ${L}_pytest.log | cut -c1-300
for c in gwc_train acv_train kitti_infer psm_volume; do timeout 700 python bench.py --config $c $( [ $c = gwc_train ] || echo --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; cut -c1-160 ${L}_bench_$c.json; done
( cd /tmp && rm -rf /tmp/prof_bench && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -14 ${L}_bench_kernel_trace_steady.txt | cut -c1-150; grep -E "cost_volume|total kernel" ${L}_bench_kernel_trace_steady.txt | cut -c1-150
( cd /tmp && rm -rf /tmp/prof_team && STX_CVB_TEAM=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_team -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /tmp/team.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_team --steady cost_volume_fwd 3 2>&1 | grep -E "cost_volume|total kernel" | cut -c1-150 > ${L}_trace_cvb_team.txt; cat ${L}_trace_cvb_team.txt
timeout 600 python tools/kernel_bench.py --iters 20 --cold > ${L}_kb_cold.log 2>&1; grep -E '"kernel"' ${L}_kb_cold.log | cut -c1-150 > ${L}_kernel_bench_cold.jsonl; wc -l ${L}_kernel_bench_cold.jsonl
