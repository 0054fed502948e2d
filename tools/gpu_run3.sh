#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|Error" gpurun_out/pytest_gpu.log | tail -8
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python tools/kernel_bench.py --iters 5 > gpurun_out/kernel_bench.log 2>&1; cat gpurun_out/kernel_bench.log
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmc_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 2 --only conv_32_32_L0 > /root/repo/gpurun_out/pmc_$tag.log 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmc_$tag conv3d > /root/repo/gpurun_out/pmc_$tag.txt 2>&1
  cat /root/repo/gpurun_out/pmc_$tag.txt
done
cd /root/repo
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench2.log 2>&1; tail -1 gpurun_out/bench2.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python /root/repo/bench.py --steps 16 --warmup 4 --no-cpu-baseline > /root/repo/gpurun_out/rocprof_bench.log 2>&1
cd /root/repo; python tools/rocprof_summary.py /tmp/prof_bench > gpurun_out/prof_bench_summary.txt 2>&1; head -50 gpurun_out/prof_bench_summary.txt
