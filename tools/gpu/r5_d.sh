#!/bin/bash
# Round 5, call D: the whole GPU suite in the driver's form (f-1 kernel tests importable now), bench lines of the volume configs
# with the same-box write ceiling.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5d
rm -f gpurun_out/parity_report.jsonl
( timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=8 2>&1 | grep -v "^  " | tail -70 ) > ${L}_pytest.log 2>&1; tail -22 ${L}_pytest.log | cut -c1-400
for c in psm_volume kitti_infer; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > ${L}_bench_$c.json; python - <<EOF2
import json
d=json.load(open("${L}_bench_$c.json"))
print("$c", d["value"], d["ms_per_step"], {k: d["roofline"].get(k) for k in ("frac","same_box_write_stream_gbs","frac_of_same_box_output_fill","avg_launch_ms")}, d.get("unpadded_135x240"))
EOF2
done
