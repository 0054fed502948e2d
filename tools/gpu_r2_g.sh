#!/bin/bash
# Round 2, GPU call G: steady-state kernel trace of the bench (new kernels), PMC of march v2 and of the cost-volume builder.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "bn" > gpurun_out/pytest_g.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_g.log | tail -4
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_g.log | cut -c1-330
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/rocprof_bench_g.log 2>&1
cd /root/repo; python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > gpurun_out/prof_bench_steady_g.txt 2>&1; head -64 gpurun_out/prof_bench_steady_g.txt | cut -c1-190
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmcm_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only conv_32_32_L0_fwd > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmcm_$tag marchw > /root/repo/gpurun_out/pmc_marchw_$tag.txt 2>&1
done
cat /root/repo/gpurun_out/pmc_marchw_*.txt | cut -c1-150
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pmcw_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only cost_volume > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pmcw_$tag cost_volume_fwd > /root/repo/gpurun_out/pmc_cvf_$tag.txt 2>&1
done
cat /root/repo/gpurun_out/pmc_cvf_*.txt | cut -c1-150
