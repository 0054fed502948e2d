#!/bin/bash
# Round 4, call P: the torch elementwise kernels of the train step split by launch grid (which gradient accumulations are volume-sized?)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rm -rf /tmp/tr && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr -o tr --output-format csv -- python $R/bench.py --config gwc_train --steps 6 --warmup 3 --no-cpu-baseline > /tmp/tr.log 2>&1 )
python tools/rocprof_summary.py /tmp/tr --steady cost_volume_fwd 3 --by-grid elementwise > gpurun_out/r4p_trace_elementwise_by_grid.txt 2>&1
grep -A60 "by launch grid" gpurun_out/r4p_trace_elementwise_by_grid.txt | cut -c1-200
