#!/bin/bash
# Round 5, call B: which op behind ACVNet's feature maps is not run-to-run reproducible (tools/acv_determinism.py)
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r5_acv_determinism.jsonl
timeout 600 python tools/acv_determinism.py 2>&1 | grep -v Warning | tail -20 | cut -c1-400
