#!/bin/bash
# Round 4, call V: BatchNorm backward-apply inside the march weight gradient -- whole GPU suite on that build, the step with and
# without it (STX_BN_BWD_IN_WGRAD), the ACVNet line, the GwcNet_GC step trace
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4v
R=$PWD
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^  " | tail -60 ) > ${L}_pytest.log 2>&1; tail -4 ${L}_pytest.log | cut -c1-300
for e in 0 1; do STX_BN_BWD_IN_WGRAD=$e timeout 300 python bench.py --config gwc_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_gwc_$e.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_gwc_$e.json"))
print("gwc_train bn_in_wgrad=$e", d["value"], d["ms_per_step"], d.get("hot_path_ms"), d.get("feature_cnn_ms"), d["roofline"]["frac"])
EOF2
done
timeout 300 python bench.py --config acv_train --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_acv_train.json; cut -c1-170 ${L}_bench_acv_train.json
( cd /tmp && rm -rf /tmp/prof_bench && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /tmp/rb.log 2>&1 )
python tools/rocprof_summary.py /tmp/prof_bench --steady cost_volume_fwd 3 > ${L}_bench_kernel_trace_steady.txt 2>&1; head -12 ${L}_bench_kernel_trace_steady.txt | cut -c1-150; grep -E "bn_|total kernel" ${L}_bench_kernel_trace_steady.txt | cut -c1-130
