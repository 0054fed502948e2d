#!/bin/bash
# Round 5, call E: the train-step tests at toy shapes (phases 3 / 4 of the GPU order) on the antithetic envelope, all of them (no -x)
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5e
( timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -k "train" --durations=8 2>&1 | grep -v "^  " | tail -60 ) > ${L}_pytest.log 2>&1; tail -32 ${L}_pytest.log | cut -c1-400
