#!/bin/bash
# Round 2, GPU call W (the last ~60 s): the benchmarked train step at full size and the CFNet / PCWNet paths, final defaults.
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( timeout 52 python -m pytest "tests/test_models.py::test_gwcnet_gc_full_size_train_step_parity" "tests/test_models.py::test_cfnet_eval_parity" "tests/test_models.py::test_sampled_volume_into_conv_autograd" "tests/test_models.py::test_pcwnet_gc_eval_parity_gpu" "tests/test_models.py::test_pcwnet_gc_train_step_gpu" -m gpu -q -p no:cacheprovider 2>&1 | tail -6 ) > gpurun_out/pytest_gpu_w.log 2>&1; cut -c1-300 gpurun_out/pytest_gpu_w.log
