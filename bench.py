#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): stereo pairs/s of one GwcNet_GC train step (fwd + bwd +
gradient all-reduce + optimizer) on synthetic 540x960 SceneFlow-shape pairs, D=192, fp32.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Shapes: the reference pads 540x960 to 576x960 (`pad_to_2x`, datasets/data_augmentation/__init__.py:57-80);
throughput is counted per original pair.  One rank per GPU, per-GPU batch fixed (weak scaling),
gradients averaged with one RCCL all-reduce over a flat bucket.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: the 3x3x3 stride-1 32->32 fp32-MFMA
implicit-GEMM Conv3d `conv3d_march_kernel<1,16,2>`; achieved = algorithmic FLOPs of its launches / their HIP-event time inside the
timed region) and `cpu_baseline` (the CPU oracle -- a torch-op restatement of the reference -- timed on
this box's host cores on a bounded sample; baseline only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md
LOSS_W = (0.5, 0.5, 0.7, 1.0)      # GwcNet paper weights (the reference ships no supervised loss)


def smooth_l1_multi(preds, gt, maxdisp):
    """The sync-free masked multi-output loss of the package (stereo_toolbox_amd/losses.py) with the GwcNet weights."""
    from stereo_toolbox_amd.losses import masked_smooth_l1_multi
    return masked_smooth_l1_multi(preds, gt, maxdisp, LOSS_W)


class KernelTimer:
    """HIP-event timing of selected C-ABI launches on the stream they are issued on."""

    def __init__(self):
        self.records = []
        self.enabled = False

    def install(self):
        from stereo_toolbox_amd import ops
        orig = ops.conv3d_forward
        timer = self

        def timed(x, wp, Cout, ks, stride, *a, **k):
            if not (timer.enabled and ks == 3 and stride == 1 and Cout <= 32 and x.shape[-1] == 32):
                return orig(x, wp, Cout, ks, stride, *a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(x, wp, Cout, ks, stride, *a, **k)
            e1.record()
            B, D, H, W, Cin = x.shape
            timer.records.append((e0, e1, 2.0 * B * D * H * W * Cout * Cin * 27))
            return out
        ops.conv3d_forward = timed

    def summary(self):
        if not self.records:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.records)
        fl = sum(f for _, _, f in self.records)
        return {"launches": len(self.records), "ms_total": ms, "flops_total": fl}


def pmc_traffic_bytes():
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, separate passes, KB per dispatch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950).  Counters cannot be collected inside this process, so this is the figure of the profiled session of the
    same kernel, or None when the summary is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_conv3d_march.txt")
    try:
        vals = {}
        with open(path) as f:
            for line in f:
                parts = line.split()
                if parts and parts[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[parts[0]] = float(line.split("avg=")[1])
        return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0)
    except Exception:
        return None


def cpu_baseline(maxdisp):
    """Oracle (torch-op restatement of the reference path) fwd+bwd on the host cores, bounded sample."""
    from oracle import torch_oracle as O
    from stereo_toolbox_amd.models import GwcNet_GC
    from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor
    H, W = 288, 480                                 # 1/4 of the 576x960 pixels
    m = GwcNet_GC(maxdisp)
    sd = m.state_dict()
    fill_state_dict(sd)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    left, right = synthetic_tensor((1, 3, H, W), 1), synthetic_tensor((1, 3, H, W), 2)
    gt = synthetic_tensor((1, H, W), 3, lo=0.0, hi=190.0)
    t0 = time.time()
    preds = O.gwcnet_forward(sd, left, right, maxdisp, True, training=True)
    O.smooth_l1_multi(preds, gt, maxdisp, LOSS_W).backward()
    dt = time.time() - t0
    frac = (H * W) / (576.0 * 960.0)
    return {"value": round(frac / dt, 5), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle GwcNet_GC fwd+bwd, 1 pair at {H}x{W} D={maxdisp} ({dt:.1f} s), scaled by pixel ratio "
                      f"{frac:.3f} to the 576x960 pair"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="pairs per GPU")
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--maxdisp", type=int, default=192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (fwd+bwd+all-reduce+Adam) in one hipGraph and replay it")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP hot path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from stereo_toolbox_amd.distributed import FlatGradSync, broadcast_parameters
    from stereo_toolbox_amd.models import GwcNet_GC
    from stereo_toolbox_amd.utils import fill_state_dict

    torch.backends.cudnn.benchmark = True
    model = GwcNet_GC(args.maxdisp)
    sd = model.state_dict()
    fill_state_dict(sd)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    broadcast_parameters(model)
    sync = FlatGradSync(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=args.graph)

    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)
    B, H, W = args.batch, args.height, args.width
    left = torch.randn(B, 3, H, W, device=dev, generator=g)
    right = torch.randn(B, 3, H, W, device=dev, generator=g)
    gt = 190.0 * torch.rand(B, H, W, device=dev, generator=g)

    timer = KernelTimer()
    timer.install()

    def step():
        sync.detach_grads()
        preds = model(left, right)
        loss = smooth_l1_multi(preds, gt, args.maxdisp)
        loss.backward()
        sync.pack()
        sync.all_reduce()
        opt.step()

    if args.graph:
        # hipGraph path: warm up on a side stream (MIOpen search, workspaces), then capture one step.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(2, args.warmup)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        sync.detach_grads()
        with torch.cuda.graph(graph):
            step()
        eager_step = step
        step = graph.replay            # noqa: F811
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = not args.graph   # per-launch HIP events cannot be recorded inside a replayed graph
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ks = timer.summary()
        roof = None
        if ks:
            ach = ks["flops_total"] / (ks["ms_total"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "conv3d_march_kernel<1,16,2> (3x3x3 stride-1 Conv3d 32->32 fwd/dgrad, fp32 MFMA)",
                    "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": pmc_traffic_bytes(),
                    "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate "
                                    "passes (profiles/r01_pmc_conv3d_march.txt); algorithmic 424 MB",
                    "launches_per_step": ks["launches"] // max(1, args.steps),
                    "avg_launch_ms": round(ks["ms_total"] / ks["launches"], 4)}
        out = {
            "metric": "stereo pairs/sec @ 540x960 D=192 (GwcNet_GC fwd+bwd)",
            "value": round(world * B * args.steps / dt, 4),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"GwcNet_GC(maxdisp={args.maxdisp}) train step: fwd+bwd+allreduce+Adam, "
                                   f"{H}x{W} (540x960 padded by pad_to_2x) pairs, batch {B}/GPU, fp32, "
                                   "synthetic randn inputs, deterministic filler weights",
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "launch": "hipGraph replay" if args.graph else "eager"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.maxdisp)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
