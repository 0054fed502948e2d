#!/bin/bash
# Round 5, call M: the whole GPU suite in the driver's own form on the final build
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5m
rm -f gpurun_out/parity_report.jsonl
( timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=12 2>&1 | grep -v "^  " | tail -90 ) > ${L}_pytest.log 2>&1; tail -24 ${L}_pytest.log | cut -c1-200
