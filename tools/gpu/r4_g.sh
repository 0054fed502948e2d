#!/bin/bash
# Round 4, call G: re-formulated tests, patch-convolution kernel after the XCD-contiguous runs (table + counters), bench lines
# (headline with the CPU baseline; events now bracket the C-ABI launch itself).
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4g
R=$PWD
( timeout 900 python -m pytest tests/test_models.py tests/test_kernels.py tests/test_trainer_dropin.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^  " | tail -40 ) > ${L}_pytest.log 2>&1; grep -E "passed|failed|^FAILED|^E  " ${L}_pytest.log | cut -c1-300
timeout 300 python tools/kernel_bench.py --iters 20 --cold --only cost_volume,dwconv > ${L}_kb_cold.log 2>&1; grep -E '"kernel"' ${L}_kb_cold.log | cut -c1-150
timeout 300 python tools/kernel_bench.py --iters 20 --only cost_volume,dwconv > ${L}_kb_warm.log 2>&1; grep -E '"kernel"' ${L}_kb_warm.log | cut -c1-150
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && rm -rf /tmp/pmc_x && timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_x -o pmc --output-format csv -- python $R/tools/kernel_bench.py --iters 3 --only dwconv > /dev/null 2>&1 )
  python tools/pmc_summary.py /tmp/pmc_x dwconv >> ${L}_pmc_dwconv.txt 2>&1
done
cut -c1-110 ${L}_pmc_dwconv.txt
for c in gwc_train acv_train psm_volume; do timeout 700 python bench.py --config $c $( [ $c = gwc_train ] || echo --no-cpu-baseline ) 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_$c.json"))
r=d["roofline"]; print("$c", d["value"], d["ms_per_step"], r["frac"], r.get("avg_launch_ms"), (d.get("roofline_volume_build") or {}).get("frac"), (d.get("roofline_volume_build") or {}).get("avg_launch_ms"))
if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["extrapolated"], d["cpu_baseline"]["sample_s"])
EOF2
done
