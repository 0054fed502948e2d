#!/bin/bash
# Round 3, GPU call H1: the COMPLETE -m gpu suite on the pruned build, then one bench line per BASELINE config.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3h1
rm -f gpurun_out/parity_report.jsonl
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 ) > ${L}_pytest.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" ${L}_pytest.log | tail -12
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -v Warning | tail -1 > ${L}_bench_gwc_train.json; cut -c1-400 ${L}_bench_gwc_train.json
timeout 600 python bench.py --config acv_train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v Warning | tail -1 > ${L}_bench_acv_train.json; cut -c1-300 ${L}_bench_acv_train.json
timeout 600 python bench.py --config kitti_infer --steps 30 --warmup 5 2>&1 | grep -v Warning | tail -1 > ${L}_bench_kitti_infer.json; cut -c1-300 ${L}_bench_kitti_infer.json
timeout 300 python bench.py --config psm_volume --steps 50 --warmup 5 2>&1 | grep -v Warning | tail -1 > ${L}_bench_psm_volume.json; cut -c1-300 ${L}_bench_psm_volume.json
mkdir -p gpurun_out/miopen_db; cp stereo_toolbox_amd/tuning/miopen/*.txt gpurun_out/miopen_db/
