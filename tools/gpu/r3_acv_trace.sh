mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_acv -o bench --output-format csv -- python $R/bench.py --config acv_train --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r3acv_rocprof.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_acv --steady cost_volume_fwd 4 > gpurun_out/r3acv_kernel_trace_steady.txt 2>&1; head -45 gpurun_out/r3acv_kernel_trace_steady.txt | cut -c1-160; tail -1 gpurun_out/r3acv_kernel_trace_steady.txt
