#!/bin/bash
# Round 5, call N: first the new exact DDP / FlatGradSync test alone, then the whole GPU suite in the driver's own form
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5n
( timeout 600 python -m pytest tests/test_trainer_dropin.py -x -q -m gpu -p no:cacheprovider -k exact 2>&1 | tail -15 ) | cut -c1-200
rm -f gpurun_out/parity_report.jsonl
( timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=12 2>&1 | grep -v "^  " | tail -90 ) > ${L}_pytest.log 2>&1; tail -24 ${L}_pytest.log | cut -c1-200
