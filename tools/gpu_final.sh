#!/bin/bash
# Final validation + profiles of the round: gpu tests, smoke, bench line, kernel table, rocprof summary, PMC of the dominant kernel.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_gpu.log | tail -8
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-1800
timeout 600 python tools/kernel_bench.py --iters 5 > gpurun_out/kernel_bench.log 2>&1; grep kernel gpurun_out/kernel_bench.log > gpurun_out/kernel_bench.jsonl; cat gpurun_out/kernel_bench.jsonl | cut -c1-110
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/rocprof_bench.log 2>&1
cd /root/repo; python tools/rocprof_summary.py /tmp/prof_bench > gpurun_out/prof_bench_summary.txt 2>&1; head -14 gpurun_out/prof_bench_summary.txt | cut -c1-150
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmcf_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 2 --only conv_32_32_L0_fwd > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmcf_$tag march > /root/repo/gpurun_out/pmc_march_$tag.txt 2>&1
done
cat /root/repo/gpurun_out/pmc_march_*.txt
