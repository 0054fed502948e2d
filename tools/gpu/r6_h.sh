#!/bin/bash
# Round 6, call H: the round's evidence on the final build -- GPU suite in the driver's form, smoke, the four BASELINE bench lines
# (headline with the CPU baseline), the step trace, fetch / write counters of the volume builder and the march forward kernel
# STAMPED with the sha256 of their sources (bench.py's `traffic` constants; a stale stamp -> null).
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r6h
R=$PWD
rm -f gpurun_out/parity_report.jsonl
( time timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=15 2>&1 | grep -v "^  " | tail -80 ) > ${L}_pytest.log 2>&1; tail -6 ${L}_pytest.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > ${L}_smoke.log 2>&1; tail -2 ${L}_smoke.log
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 5 --only cost_volume > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x cost_volume_fwd >> ${L}_pmc_cost_volume_fwd.txt 2>&1
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only conv_32_32_L0_fwd > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x marchw >> ${L}_pmc_conv3d_marchw.txt 2>&1
done
python -c "import bench; print('kernel_source_sha', bench.kernel_source_sha('cost_volume_mfma.hip'))" >> ${L}_pmc_cost_volume_fwd.txt
python -c "import bench; print('kernel_source_sha', bench.kernel_source_sha('conv3d.hip'))" >> ${L}_pmc_conv3d_marchw.txt
cat ${L}_pmc_cost_volume_fwd.txt ${L}_pmc_conv3d_marchw.txt | cut -c1-120
mkdir -p profiles; cp ${L}_pmc_cost_volume_fwd.txt profiles/r06_pmc_cost_volume_fwd.txt; cp ${L}_pmc_conv3d_marchw.txt profiles/r06_pmc_conv3d_marchw.txt   # (read by the bench lines below)
for c in gwc_train acv_train kitti_infer psm_volume; do timeout 700 python bench.py --config $c $( [ $c = gwc_train ] || echo --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; cut -c1-170 ${L}_bench_$c.json; done
( cd /tmp && rm -rf /tmp/prof_bench && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -8 ${L}_bench_kernel_trace_steady.txt | cut -c1-150
( cd /tmp && rm -rf /tmp/prof_acv && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_acv -o bench --output-format csv -- python $R/bench.py --config acv_train --steps 4 --warmup 2 --no-cpu-baseline > /tmp/acv.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_acv --steady cost_volume_fwd 4 > ${L}_acv_train_kernel_trace_steady.txt 2>&1; head -6 ${L}_acv_train_kernel_trace_steady.txt | cut -c1-150
timeout 400 python tools/kernel_bench.py --cold --iters 10 > ${L}_kernel_bench_cold.jsonl 2>/dev/null; wc -l ${L}_kernel_bench_cold.jsonl
