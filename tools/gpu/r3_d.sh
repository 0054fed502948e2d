#!/bin/bash
# Round 3, GPU call D: (1) MFMA tail-read micro-test (why do the blocked-sum variants fail on the chip?), (2) is the step
# host-bound with the fused 2-D glue (hipGraph replay vs eager), (3) per-kernel split of the fused extractor.
mkdir -p gpurun_out
export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=$PWD/stereo_toolbox_amd/tuning/miopen
L=gpurun_out/r3d
timeout 60 tools/ubench/mfma_tail_read 2>&1 | tee ${L}_mfma_tail_read.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v Warning | tail -1 | cut -c1-330 | sed "s/^/eager /" | tee -a ${L}_bench.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --graph 2>&1 | grep -v Warning | tail -1 | cut -c1-330 | sed "s/^/graph /" | tee -a ${L}_bench.txt
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_feat2d_fused -o t -- python $R/tools/feat2d_bench.py --fmt nchw --iters 5 --no-eval > /dev/null 2>&1 )
python tools/rocprof_summary.py gpurun_out/prof_feat2d_fused 2>&1 | head -40 | cut -c1-150 | tee ${L}_feat2d_fused_kernels.txt
rm -rf gpurun_out/prof_feat2d_fused
