#!/bin/bash
# Round 5, call T: bench.py under the driver's multi-GPU launcher form at one rank (torch.distributed.run, RCCL backend), and the
# two loosened smoke tests
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | cut -c1-400 | tee gpurun_out/r5t_bench_torchrun_1rank.json
( timeout 600 python -m pytest tests/test_trainer_dropin.py tests/test_models.py -q -m gpu -p no:cacheprovider -k "ddp or flat_grad or frozen" 2>&1 | tail -3 ) | cut -c1-200
