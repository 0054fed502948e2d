#!/bin/bash
# Round 5, call L: evidence on the final build -- the whole GPU suite in the driver's own form, the four BASELINE bench lines (headline
# with the CPU baseline), GwcNet_GC and ACVNet step traces, the cold kernel table.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5l
R=$PWD
rm -f gpurun_out/parity_report.jsonl
( timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=12 2>&1 | grep -v "^  " | tail -90 ) > ${L}_pytest.log 2>&1; tail -5 ${L}_pytest.log | cut -c1-300
for c in gwc_train acv_train kitti_infer psm_volume; do timeout 700 python bench.py --config $c $( [ $c = gwc_train ] || echo --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; cut -c1-170 ${L}_bench_$c.json; done
( cd /tmp && rm -rf /tmp/prof_bench && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -8 ${L}_bench_kernel_trace_steady.txt | cut -c1-150; grep -E "cost_volume|total kernel" ${L}_bench_kernel_trace_steady.txt | cut -c1-150
( cd /tmp && rm -rf /tmp/prof_acv && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_acv -o bench --output-format csv -- python $R/bench.py --config acv_train --steps 4 --warmup 2 --no-cpu-baseline > /tmp/acv.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_acv --steady cost_volume_fwd 4 > ${L}_acv_train_kernel_trace_steady.txt 2>&1; head -6 ${L}_acv_train_kernel_trace_steady.txt | cut -c1-150
timeout 400 python tools/kernel_bench.py --cold --iters 10 > ${L}_kernel_bench_cold.jsonl 2>/dev/null; wc -l ${L}_kernel_bench_cold.jsonl
