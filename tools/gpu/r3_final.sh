#!/bin/bash
# Round 3, last GPU call: evidence refresh on the final build (whole GPU suite, kernel table + A/B sets, bench lines, step trace,
# PMC of the march kernel).
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3final
R=$PWD
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 ) > ${L}_pytest.log 2>&1; tail -2 ${L}_pytest.log | cut -c1-200
timeout 400 python tools/kernel_bench.py --iters 20 --ab > ${L}_kernel_bench.log 2>&1; grep -E '"kernel"|"ab"' ${L}_kernel_bench.log > ${L}_kernel_bench.jsonl; wc -l ${L}_kernel_bench.jsonl
for c in gwc_train acv_train kitti_infer psm_volume; do timeout 600 python bench.py --config $c $( [ $c = gwc_train ] || echo --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; cut -c1-200 ${L}_bench_$c.json; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -8 ${L}_bench_kernel_trace_steady.txt | cut -c1-150
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only conv_32_32_L0_fwd > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x marchw >> ${L}_pmc_conv3d_marchw.txt 2>&1
done
cut -c1-110 ${L}_pmc_conv3d_marchw.txt
for m in GwcNet_GC; do for s in "480 640" "736 1280" "1088 1920"; do set -- $s; timeout 300 python tools/speed_test.py --model $m --height $1 --width $2 --warmup 5 --iters 30 2>&1 | grep '^{' | tee -a ${L}_inference_speed.jsonl | cut -c1-120; done; done
