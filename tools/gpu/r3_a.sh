#!/bin/bash
# Round 3, GPU call A: (1) time what round 2 left untimed, (2) attribute the full-size eval error to the 2-D CNN / the HIP
# path, (3) 2-D CNN NCHW vs channels_last, (4) capture MIOpen's find results for the bench shapes, (5) wgrad ablations.
mkdir -p gpurun_out/miopen_db
export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
L=gpurun_out/r3a
( timeout 200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "sampled or stride2_dense or cost_volume_fwd_bwd" 2>&1 | tail -4 ) > ${L}_pytest.log 2>&1; cat ${L}_pytest.log
timeout 240 python tools/kernel_bench.py --iters 10 --ab > ${L}_kernel_bench.log 2>&1; grep -E '"kernel"|"ab"' ${L}_kernel_bench.log > ${L}_kernel_bench.jsonl; cut -c1-120 ${L}_kernel_bench.jsonl | head -60
for ab in 1 2; do STX_WGRAD_ABLATE=$ab timeout 60 python tools/kernel_bench.py --iters 10 --only conv_32_32_L0_wgrad,conv_64_64_L1_wgrad,conv_32_64_s2_L0_wgrad 2>&1 | grep kernel | sed "s/^/wgrad_ablate=$ab /" | tee -a ${L}_wgrad_ablate.txt; done
rm -f gpurun_out/parity_isolation.jsonl
timeout 300 python tools/parity_isolation.py --tag gwc_gc_384x1248 --label default 2>&1 | tail -1 | cut -c1-1500
MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout 300 python tools/parity_isolation.py --tag gwc_gc_384x1248 --label no_winograd 2>&1 | tail -1 | cut -c1-600
timeout 400 python tools/feat2d_bench.py 2>&1 | grep -v Warning | tee ${L}_feat2d.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v Warning | tail -2 | tee ${L}_bench.txt
ls -la gpurun_out/miopen_db | head; du -sh ~/.cache/miopen ~/.config/miopen 2>/dev/null
# per-kernel split of the 2-D CNN (find results are in the user db by now: no search in the trace)
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_feat2d_nchw -o t -- python $R/tools/feat2d_bench.py --fmt nchw --iters 5 --no-eval > /dev/null 2>&1 )
python tools/rocprof_summary.py gpurun_out/prof_feat2d_nchw 2>&1 | head -45 | cut -c1-170 | tee ${L}_feat2d_nchw_kernels.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_feat2d_nhwc -o t -- python $R/tools/feat2d_bench.py --fmt nhwc --iters 5 --no-eval > /dev/null 2>&1 )
python tools/rocprof_summary.py gpurun_out/prof_feat2d_nhwc 2>&1 | head -45 | cut -c1-170 | tee ${L}_feat2d_nhwc_kernels.txt
rm -rf gpurun_out/prof_feat2d_nchw gpurun_out/prof_feat2d_nhwc
