#!/bin/bash
# Round 5, call S: the whole GPU suite in the driver's own form after the last TEST-side change (envelope reach); product unchanged since call N
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5s
rm -f gpurun_out/parity_report.jsonl
( timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=8 2>&1 | grep -v "^  " | tail -80 ) > ${L}_pytest.log 2>&1; tail -18 ${L}_pytest.log | cut -c1-260
