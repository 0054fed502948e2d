#!/bin/bash
# Round 4, call S: convolutions that read one volume share an autograd node (input gradients accumulated in kernel epilogues):
# parity, step time, the volume-sized adds left in the trace
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4s
R=$PWD
( timeout 900 python -m pytest tests/test_models.py tests/test_trainer_dropin.py -m gpu -q -p no:cacheprovider -k "gwcnet or trainer or ddp or flat or cfnet or pcwnet" 2>&1 | grep -v "^  " | tail -8 ) > ${L}_pytest.log 2>&1; grep -E "passed|failed|^FAILED|^E  " ${L}_pytest.log | cut -c1-300
timeout 400 python bench.py --config gwc_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_gwc.json"))
print("gwc_train", d["value"], d["ms_per_step"], d.get("hot_path_ms"), d.get("feature_cnn_ms"), d["roofline"]["frac"])
EOF2
( cd /tmp && rm -rf /tmp/tr && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr -o tr --output-format csv -- python $R/bench.py --config gwc_train --steps 6 --warmup 3 --no-cpu-baseline > /tmp/tr.log 2>&1 )
python tools/rocprof_summary.py /tmp/tr --steady cost_volume_fwd 3 --by-grid "CUDAFunctor_add" > ${L}_trace.txt 2>&1
head -34 ${L}_trace.txt | cut -c1-150; grep -A12 "by launch grid" ${L}_trace.txt | cut -c1-150; grep "total kernel" ${L}_trace.txt
