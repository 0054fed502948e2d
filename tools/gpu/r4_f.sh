#!/bin/bash
# Round 4, call F: evidence on the current build -- whole GPU suite, the four BASELINE bench lines (headline with the CPU
# baseline), step traces of the GwcNet_GC and ACVNet train steps, counters of the ACVNet-only kernels, cost-volume / patch-conv
# kernel table (cold + warm), IGEV aggregation timing.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4f
R=$PWD
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^  " | tail -60 ) > ${L}_pytest.log 2>&1; tail -4 ${L}_pytest.log | cut -c1-300
for c in gwc_train acv_train kitti_infer psm_volume; do timeout 700 python bench.py --config $c $( [ $c = gwc_train ] || echo --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; cut -c1-160 ${L}_bench_$c.json; done
( cd /tmp && rm -rf /tmp/prof_bench && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -14 ${L}_bench_kernel_trace_steady.txt | cut -c1-150
( cd /tmp && rm -rf /tmp/prof_acv && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_acv -o bench --output-format csv -- python $R/bench.py --config acv_train --steps 4 --warmup 2 --no-cpu-baseline > $R/${L}_rocprof_acv.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_acv --steady cost_volume_fwd 4 > ${L}_acv_train_kernel_trace_steady.txt 2>&1; head -12 ${L}_acv_train_kernel_trace_steady.txt | cut -c1-150
timeout 300 python tools/kernel_bench.py --iters 20 --cold --only cost_volume,dwconv > ${L}_kb_cold.log 2>&1; grep -E '"kernel"' ${L}_kb_cold.log | cut -c1-150
timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume,dwconv > ${L}_kb_warm.log 2>&1; grep -E '"kernel"' ${L}_kb_warm.log | cut -c1-150
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only dwconv > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x dwconv >> ${L}_pmc_dwconv.txt 2>&1
done
cut -c1-110 ${L}_pmc_dwconv.txt
