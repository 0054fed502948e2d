#!/bin/bash
# Round 6, call R: the whole GPU suite in the driver's form on the build with csrc/conv2d.hip, smoke(), the bench line (with the CPU
# baseline), the other configs' lines and the step's kernel trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r6r
R=$PWD
rm -f gpurun_out/parity_report.jsonl
(time python -m pytest tests/ -x -q -m gpu --durations=15) > ${L}_pytest.log 2>&1; echo rc=$? >> ${L}_pytest.log
cp gpurun_out/parity_report.jsonl ${L}_parity_report.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > ${L}_smoke.log 2>&1; echo rc=$? >> ${L}_smoke.log
timeout 700 python bench.py 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc_train.json; cut -c1-170 ${L}_bench_gwc_train.json
for c in acv_train kitti_infer psm_volume; do timeout 400 python bench.py --config $c --no-cpu-baseline 2>&1 | grep '^{' | tail -1 >> ${L}_bench_configs.jsonl; done
( cd /tmp && rm -rf /tmp/prof_bench && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -8 ${L}_bench_kernel_trace_steady.txt | cut -c1-150
tail -4 ${L}_pytest.log; tail -2 ${L}_smoke.log; cut -c1-200 ${L}_bench_configs.jsonl
