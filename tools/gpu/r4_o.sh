#!/bin/bash
# Round 4, call O: BatchNorm streaming kernels with several positions per thread in flight: parity + cold table + step time
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4o
( timeout 600 python -m pytest tests/test_kernels.py tests/test_igev_aggregation.py -m gpu -q -p no:cacheprovider -k "bn or igev" 2>&1 | grep -v "^  " | tail -8 ) > ${L}_pytest.log 2>&1; grep -E "passed|failed|^FAILED|^E  " ${L}_pytest.log | cut -c1-300
timeout 300 python tools/kernel_bench.py --iters 30 --cold --only bn > ${L}_kb_cold.log 2>&1; grep -E '"kernel"' ${L}_kb_cold.log | cut -c1-200
timeout 400 python bench.py --config gwc_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc_train.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_gwc_train.json"))
r=d["roofline"]; print("gwc_train", d["value"], d["ms_per_step"], r["frac"], d.get("hot_path_ms"), d.get("feature_cnn_ms"))
EOF2
R=$PWD
( cd /tmp && rm -rf /tmp/tr && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr -o tr --output-format csv -- python $R/bench.py --config gwc_train --steps 6 --warmup 3 --no-cpu-baseline > /tmp/tr.log 2>&1 )
python tools/rocprof_summary.py /tmp/tr --steady cost_volume_fwd 3 --by-grid bn_ > ${L}_trace_bn_by_grid.txt 2>&1
grep -E "bn_|total kernel" ${L}_trace_bn_by_grid.txt | cut -c1-140
