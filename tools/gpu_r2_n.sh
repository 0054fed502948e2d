#!/bin/bash
# Round 2, GPU call N: validation of HEAD in ONE call, most important artefacts first, every stage skipped once the
# call's own clock passes its budget (the pool charges box time whether or not the command uses it).
#   1 GPU test-suite (3 xdist workers: the full-size tests run the CPU oracle on the host cores) + parity report
#   2 headline bench line      3 kernel table      4 cost-volume backward A/B (new MFMA kernel / first generation / schedules)
#   5 steady-state kernel trace of the bench      6 PMC of the cost-volume backward      7 the other BASELINE configs
mkdir -p gpurun_out
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
left() { [ $(el) -lt $1 ]; }
rm -f gpurun_out/parity_report.jsonl
( timeout 540 python -m pytest tests -m gpu -q -p no:cacheprovider -n 3 > gpurun_out/pytest_gpu_n.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_n.log )
grep -E "passed|failed|^FAILED|^ERROR|rc=" gpurun_out/pytest_gpu_n.log | tail -12
echo "[t=$(el)s] pytest done"
timeout 300 python bench.py > gpurun_out/bench_n.log 2>&1; tail -1 gpurun_out/bench_n.log | cut -c1-2600
echo "[t=$(el)s] bench done"
if left 560; then
  timeout 240 python tools/kernel_bench.py --iters 10 > gpurun_out/kernel_bench_n.log 2>&1; grep kernel gpurun_out/kernel_bench_n.log > gpurun_out/kernel_bench_n.jsonl; cut -c1-120 gpurun_out/kernel_bench_n.jsonl
  echo "[t=$(el)s] kernel table done"
fi
if left 600; then
  for v in "STX_CVB_OLD=1" "STX_CVB_TEAM=1" "STX_CVB_TEAM=0" "STX_CVB_TEAM=1 STX_CVB_NSET=2" "STX_CVB_TEAM=1 STX_CVB_NSET=4"; do
    echo "== cost volume bwd [$v]" | tee -a gpurun_out/cvb_ab_n.log
    env $v timeout 120 python tools/kernel_bench.py --iters 20 --only cost_volume 2>&1 | grep -E "kernel.*bwd" | tee -a gpurun_out/cvb_ab_n.log | cut -c1-150
  done
  for v in "STX_MARCH_6464=0" "STX_MARCH_6464=1"; do
    echo "== 64->64 L1 [$v]" | tee -a gpurun_out/cvb_ab_n.log
    env $v timeout 120 python tools/kernel_bench.py --iters 20 --only conv_64_64_L1_fwd 2>&1 | grep -E "kernel" | tee -a gpurun_out/cvb_ab_n.log | cut -c1-150
  done
  echo "[t=$(el)s] A/B done"
fi
if left 640; then
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/rocprof_bench_n.log 2>&1 )
  python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > gpurun_out/prof_bench_steady_n.txt 2>&1; head -40 gpurun_out/prof_bench_steady_n.txt | cut -c1-170
  echo "[t=$(el)s] trace done"
fi
if left 700; then
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | cut -d' ' -f1)
    ( cd /tmp && timeout 120 rocprofv3 --pmc $grp -d /tmp/pmcn_$tag -o pmc --output-format csv -- python /root/repo/tools/kernel_bench.py --iters 3 --only cost_volume > /dev/null 2>&1 )
    python tools/pmc_summary.py /tmp/pmcn_$tag cost_volume >> gpurun_out/pmc_cv_n.txt 2>&1
    left 760 || break
  done
  cut -c1-150 gpurun_out/pmc_cv_n.txt
  echo "[t=$(el)s] pmc done"
fi
for c in psm_volume kitti_infer acv_train; do
  left 760 || break
  timeout 200 python bench.py --config $c --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/bench_configs_n.log | cut -c1-700
done
echo "[t=$(el)s] end"
