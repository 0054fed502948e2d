#!/bin/bash
# Round 4, call E: the five tests that failed in call D with their messages; error attribution of the hand-written path's
# small-shape gradient ratio (default build vs -DSTX_PRECISE_MATH vs precise + -ffp-contract=off); IGEV aggregation timing.
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4e
( timeout 900 python -m pytest tests/test_trainer_dropin.py "tests/test_models.py::test_acvnet_train_parity" "tests/test_models.py::test_gwcnet_gc_train_parity" tests/test_models.py::test_gwcnet_gc_train_grads_hand_written_path_isolated -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^  " | tail -60 ) > ${L}_pytest.log 2>&1; grep -E "^E  |passed|failed|FAILED" ${L}_pytest.log | cut -c1-500
for lib in "" stereo_toolbox_amd/lib/libstx_hip_precise.so stereo_toolbox_amd/lib/libstx_hip_precise_nocontract.so; do
  echo "== STX_HIP_LIB=$lib"
  for rep in 1 2; do STX_HIP_LIB=$lib timeout 300 python -m pytest tests/test_models.py::test_gwcnet_gc_train_grads_hand_written_path_isolated "tests/test_models.py::test_gwcnet_gc_train_parity" "tests/test_models.py::test_acvnet_train_parity" -m gpu -q -p no:cacheprovider 2>&1 | grep -E '^\{"test"' | cut -c1-260; done
done > ${L}_attribution.txt 2>&1
cat ${L}_attribution.txt
timeout 300 python tools/igev_agg_bench.py > ${L}_igev.json 2>&1; tail -1 ${L}_igev.json
