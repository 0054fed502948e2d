#!/bin/bash
# Round 6, call I: the volume builder with windowed assignment (STX_CV_WIN) against the one-window launch: alternating A/B, cold and
# warm, and inside the train step.
mkdir -p gpurun_out
L=gpurun_out/r6i
for rep in 1 2 3; do
  python tools/kernel_bench.py --cold --iters 20 --only cost_volume_fwd --ab --ab-filter "cost volume fwd: windows" 2>/dev/null | cut -c1-260 >> ${L}_cv_win_cold.jsonl
done
python tools/kernel_bench.py --iters 20 --only cost_volume_fwd --ab --ab-filter "cost volume fwd: windows" 2>/dev/null | cut -c1-260 > ${L}_cv_win_warm.jsonl
for w in 0 256 512 0 256 512; do STX_CV_WIN=$w timeout 300 python bench.py --no-cpu-baseline --steps 20 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline_volume_build']
print(json.dumps({'STX_CV_WIN': $w, 'ms_per_step': d['ms_per_step'], 'volume_build_ms': r['avg_launch_ms'], 'frac': r['frac'], 'fill_ms': r['same_box_output_fill_ms']}))" >> ${L}_cv_win_step.jsonl; done
cat ${L}_cv_win_step.jsonl; grep -c . ${L}_cv_win_cold.jsonl
