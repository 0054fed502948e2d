#!/bin/bash
# Round 4, call R: group-total in the BN column sums (parity + step), torch glue kernels of the ACVNet train step and of the KITTI
# inference forward split by launch grid
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4r
R=$PWD
( timeout 900 python -m pytest tests/test_kernels.py tests/test_models.py -m gpu -q -p no:cacheprovider -k "bn or gwcnet_gc_train or acvnet_train" 2>&1 | grep -v "^  " | tail -8 ) > ${L}_pytest.log 2>&1; grep -E "passed|failed|^FAILED|^E  " ${L}_pytest.log | cut -c1-300
timeout 400 python bench.py --config gwc_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_gwc.json"))
print("gwc_train", d["value"], d["ms_per_step"], d.get("hot_path_ms"), d.get("feature_cnn_ms"))
EOF2
( cd /tmp && rm -rf /tmp/tr && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr -o tr --output-format csv -- python $R/bench.py --config acv_train --steps 4 --warmup 2 --no-cpu-baseline > /tmp/tr.log 2>&1 )
python tools/rocprof_summary.py /tmp/tr --steady cost_volume_fwd 4 --by-grid "at::native" > ${L}_trace_acv_glue.txt 2>&1
grep -A200 "by launch grid" ${L}_trace_acv_glue.txt | awk '$3+0 > 20' | cut -c1-170
( cd /tmp && rm -rf /tmp/tr2 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr2 -o tr --output-format csv -- python $R/bench.py --config kitti_infer --steps 6 --warmup 3 --no-cpu-baseline > /tmp/tr2.log 2>&1 )
python tools/rocprof_summary.py /tmp/tr2 --steady cost_volume_fwd 3 --by-grid "at::native" > ${L}_trace_kitti_glue.txt 2>&1
head -30 ${L}_trace_kitti_glue.txt | cut -c1-150; grep -A200 "by launch grid" ${L}_trace_kitti_glue.txt | awk '$3+0 > 8' | cut -c1-170
