#!/bin/bash
# Round 4, call Q: the feature extractors' channel concatenation as one HIP pass each way: parity + step time + copies by grid
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4q
R=$PWD
( timeout 900 python -m pytest tests/test_kernels.py tests/test_models.py -m gpu -q -p no:cacheprovider -k "concat_split or cat_features or transpose or channel_major or gwcnet or acvnet or cfnet or psmnet" 2>&1 | grep -v "^  " | tail -8 ) > ${L}_pytest.log 2>&1; grep -E "passed|failed|^FAILED|^E  " ${L}_pytest.log | cut -c1-300
for e in 1 0; do STX_FEAT2D_FUSED_CAT=$e timeout 400 python bench.py --config gwc_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_$e.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_$e.json"))
print("gwc_train cat=$e", d["value"], d["ms_per_step"], d.get("hot_path_ms"), d.get("feature_cnn_ms"))
EOF2
done
( cd /tmp && rm -rf /tmp/tr && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr -o tr --output-format csv -- python $R/bench.py --config gwc_train --steps 6 --warmup 3 --no-cpu-baseline > /tmp/tr.log 2>&1 )
python tools/rocprof_summary.py /tmp/tr --steady cost_volume_fwd 3 --by-grid elementwise > ${L}_trace_elementwise_by_grid.txt 2>&1
grep -E "cat_channels|total kernel" ${L}_trace_elementwise_by_grid.txt | cut -c1-160; grep -A14 "by launch grid" ${L}_trace_elementwise_by_grid.txt | cut -c1-150
