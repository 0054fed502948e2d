#!/bin/bash
# Round 3, GPU call H2: evidence for profiles/ on the FINAL build -- kernel table + A/B sets, steady-state kernel trace of the
# bench step, PMC (SQ set, LDS set, FETCH_SIZE, WRITE_SIZE in separate passes) of the dominant kernels, inference table.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3h2
R=$PWD
for p in 1 0; do ( STX_FEAT2D_PAIRED=$p timeout 300 python -m pytest tests/test_models.py -m gpu -q -p no:cacheprovider --tb=short -k "gwcnet_gc_train_parity or acvnet_train_parity" 2>&1 | grep -E "grad err|passed|failed|Error" | cut -c1-300 | sed "s/^/paired=$p /" ) | tee -a ${L}_small_train.txt; done
timeout 400 python tools/kernel_bench.py --iters 20 --ab > ${L}_kernel_bench.log 2>&1; grep -E '"kernel"|"ab"' ${L}_kernel_bench.log > ${L}_kernel_bench.jsonl; cut -c1-110 ${L}_kernel_bench.jsonl | head -50
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/${L}_rocprof_bench.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -70 ${L}_bench_kernel_trace_steady.txt | cut -c1-170
pmc() {   # $1 = kernel_bench filter, $2 = kernel-name substring, $3 = output tag
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $grp | cut -d' ' -f1)
    ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only $1 > /dev/null 2>&1 )
    python tools/pmc_summary.py /tmp/pmc_x $2 >> ${L}_pmc_$3.txt 2>&1
  done
  cut -c1-120 ${L}_pmc_$3.txt
}
pmc conv_32_32_L0_fwd marchw conv3d_marchw
pmc cost_volume cost_volume cost_volume
pmc conv_32_64_s2_L0_fwd,deconv_64_32_L1_fwd,conv_32_32_L0_wgrad,conv_64_64_L1_fwd conv convs
for m in PSMNet GwcNet_GC ACVNet; do for s in "480 640" "736 1280" "1088 1920"; do set -- $s; timeout 300 python tools/speed_test.py --model $m --height $1 --width $2 --warmup 5 --iters 30 2>&1 | grep '^{' | tee -a ${L}_inference_speed.jsonl; done; done
for b in "" "--batched"; do timeout 400 python tools/feat2d_bench.py --fmt nchw --no-eval $b 2>&1 | grep "fwd+bwd" | tee -a ${L}_feat2d_batched.txt; done
for c in gwc_train acv_train kitti_infer psm_volume; do timeout 600 python bench.py --config $c $( [ $c = gwc_train ] || echo --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; cut -c1-330 ${L}_bench_$c.json; done
mkdir -p gpurun_out/miopen_db; cp stereo_toolbox_amd/tuning/miopen/*.txt gpurun_out/miopen_db/
