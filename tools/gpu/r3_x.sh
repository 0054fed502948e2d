#!/bin/bash
# Round 3, GPU call X: evidence for profiles/ on the final build -- the whole GPU suite, kernel table + A/B sets, the four bench
# configs, steady-state kernel trace of the bench step, PMC (SQ set, LDS set, FETCH_SIZE, WRITE_SIZE in separate passes) of the
# dominant kernels, inference table at the reference's published resolutions.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3x
R=$PWD
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 ) > ${L}_pytest.log 2>&1; tail -3 ${L}_pytest.log | cut -c1-200
timeout 400 python tools/kernel_bench.py --iters 20 --ab > ${L}_kernel_bench.log 2>&1; grep -E '"kernel"|"ab"' ${L}_kernel_bench.log > ${L}_kernel_bench.jsonl; wc -l ${L}_kernel_bench.jsonl
for c in gwc_train acv_train kitti_infer psm_volume; do timeout 600 python bench.py --config $c $( [ $c = gwc_train ] || echo --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; cut -c1-200 ${L}_bench_$c.json; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -12 ${L}_bench_kernel_trace_steady.txt | cut -c1-150
pmc() {   # $1 = kernel_bench filter, $2 = kernel-name substring, $3 = output tag
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only $1 > /dev/null 2>&1 )
    python tools/pmc_summary.py /tmp/pmc_x "$2" >> ${L}_pmc_$3.txt 2>&1
    [ -n "$4" ] && python tools/pmc_summary.py /tmp/pmc_x "$4" >> ${L}_pmc_$5.txt 2>&1
  done
  cut -c1-120 ${L}_pmc_$3.txt | head -20
}
pmc conv_32_32_L0_fwd marchw conv3d_marchw
pmc cost_volume cost_volume cost_volume "cost_volume_fwd_mfma_kernel<8, 1, 10" cost_volume_fwd
pmc conv_32_64_s2_L0_fwd,deconv_64_32_L1_fwd,conv_32_32_L0_wgrad,conv_64_64_L1_fwd,conv_64_128_s2_L1_fwd conv convs
for m in PSMNet GwcNet_GC ACVNet; do for s in "480 640" "736 1280" "1088 1920"; do set -- $s; timeout 300 python tools/speed_test.py --model $m --height $1 --width $2 --warmup 5 --iters 30 2>&1 | grep '^{' | tee -a ${L}_inference_speed.jsonl | cut -c1-120; done; done
mkdir -p gpurun_out/miopen_db; cp stereo_toolbox_amd/tuning/miopen/*.txt gpurun_out/miopen_db/
