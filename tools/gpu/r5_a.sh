#!/bin/bash
# Round 5, call A: the whole GPU suite in the driver's own form (-x, new phase order, sensitivity-envelope train tests, the ACVNet
# isolated hand-written-path test run twice) + durations of the slowest tests.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5a
( timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=15 2>&1 | grep -v "^  " | tail -90 ) > ${L}_pytest.log 2>&1; tail -30 ${L}_pytest.log | cut -c1-400
