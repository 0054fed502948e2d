#!/bin/bash
# Round 5, call V: the whole GPU suite in the driver's form on the build with the rolling-window patch convolutions, the ACVNet
# bench line and step trace of that build, the headline line
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5v
R=$PWD
rm -f gpurun_out/parity_report.jsonl
( timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=6 2>&1 | grep -v "^  " | tail -80 ) > ${L}_pytest.log 2>&1; tail -12 ${L}_pytest.log | cut -c1-200
for c in acv_train gwc_train; do timeout 400 python bench.py --config $c --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; cut -c1-170 ${L}_bench_$c.json; done
( cd /tmp && rm -rf /tmp/prof_acv && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_acv -o bench --output-format csv -- python $R/bench.py --config acv_train --steps 4 --warmup 2 --no-cpu-baseline > /tmp/acv.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_acv --steady cost_volume_fwd 4 > ${L}_acv_train_kernel_trace_steady.txt 2>&1; grep -E "dwconv|total kernel" ${L}_acv_train_kernel_trace_steady.txt | cut -c1-150
