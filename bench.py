#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): stereo pairs/s of one GwcNet_GC train step (fwd + bwd +
gradient all-reduce + optimizer) on synthetic 540x960 SceneFlow-shape pairs, D=192, fp32.

  python bench.py --gpus N --steps K --warmup W [--config gwc_train|acv_train|kitti_infer|psm_volume]

Launch.  One rank per GPU.  Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks come
from the environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Started plainly with `--gpus N` (N > 1,
no WORLD_SIZE in the environment) bench.py re-executes itself through torch.distributed.run with N ranks
on 127.0.0.1 (reference: `torchrun --nproc_per_node=N`, trainer/trainer_torchrun.py:31-38,67-83); it refuses
to run when fewer than N devices are visible or when WORLD_SIZE disagrees with --gpus.

Configs (BASELINE.json `configs`):
  gwc_train    [2] GwcNet_GC(192) train step, 576x960 (= 540x960 after the reference's pad_to_2x,
                   datasets/data_augmentation/__init__.py:57-80), batch 1/GPU            <- the headline metric
  acv_train    [3] ACVNet(192) train step, 576x960, batch 2/GPU (global batch 16 on 8 GPUs)
  kitti_infer  [4] GwcNet_GC(192) inference, 384x1248 (= 375x1242 padded), batch-parallel (one pair per rank per step)
  psm_volume   [1] PSMNet concat cost volume 576x960 D=192 forward only (HBM roofline)
Per-GPU work is fixed (weak scaling); gradients are averaged over ranks through one flat fp32 buffer
(stereo_toolbox_amd/distributed.py), by default in 2 ranges that overlap with the backward pass.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel of the config; achieved = algorithmic FLOPs or
bytes of its launches / their HIP-event time inside the timed region, on the launching stream) and, at N=1,
`cpu_baseline` (the CPU oracle -- a torch-op restatement of the reference -- on this box's host cores, bounded
sample, warm-up + 2 timed runs; baseline only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0              # same guide (spec; a float4 copy reaches 6.29 TB/s)
LOSS_W = (0.5, 0.5, 0.7, 1.0)      # GwcNet paper weights (the reference ships no supervised loss); ACVNet: 4 outputs too

CONFIGS = {
    #               model        mode     H     W   batch  metric
    "gwc_train":   ("GwcNet_GC", "train", 576, 960, 1, "stereo pairs/sec @ 540x960 D=192 (GwcNet_GC fwd+bwd)"),
    "acv_train":   ("ACVNet", "train", 576, 960, 2, "stereo pairs/sec @ 540x960 D=192 (ACVNet fwd+bwd)"),
    "kitti_infer": ("GwcNet_GC", "eval", 384, 1248, 1, "stereo pairs/sec @ 1242x375 D=192 (GwcNet_GC inference)"),
    "psm_volume":  ("PSMNet concat volume", "volume", 576, 960, 1, "cost volumes/sec @ 540x960 D=192 (PSMNet concat volume fwd)"),
}


def smooth_l1_multi(preds, gt, maxdisp):
    """The sync-free masked multi-output loss of the package (stereo_toolbox_amd/losses.py) with the GwcNet weights."""
    from stereo_toolbox_amd.losses import masked_smooth_l1_multi
    return masked_smooth_l1_multi(preds, gt, maxdisp, LOSS_W)


class PathSplit:
    """Where a step's GPU time goes, by HIP events on the launching stream (reference template:
    evaluation/speed_and_memory_test.py:57-75 times the whole forward; here the forward and the backward are cut at the
    boundary between the 2-D feature CNN (MIOpen convolutions; since round 6 its 39 32 / 64-channel 3x3
    stride-1 layers run forward and data gradient on csrc/conv2d.hip in train mode) and the hand-written hot path behind it).  Forward: events around `model.aggregate`; backward: a tensor hook on every feature
    map handed to `aggregate` fires when its gradient exists, i.e. when the hot path's backward is done and the 2-D
    CNN's is about to start -- the last such event of a step is the cut."""
    KEYS = ("feature_cnn_fwd", "hot_path_fwd", "loss", "hot_path_bwd", "feature_cnn_bwd", "grad_sync_optimizer")

    def __init__(self, model):
        self.enabled = False
        self.steps = []
        self.cur = None
        self.model = model
        orig = model.aggregate
        split = self

        def aggregate(fl, fr, *a, **k):
            if not split.enabled:
                return orig(fl, fr, *a, **k)
            split.mark("feat_fwd_end")
            for feats in (fl, fr):
                for t in (feats.values() if isinstance(feats, dict) else [feats]):
                    if isinstance(t, torch.Tensor) and t.requires_grad:
                        t.register_hook(lambda g: (split.mark("hot_bwd_end"), None)[1])
            out = orig(fl, fr, *a, **k)
            split.mark("hot_fwd_end")
            return out
        model.aggregate = aggregate

    def mark(self, name):
        if self.enabled and self.cur is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.cur[name] = e                      # (a later mark of the same name replaces the earlier one)

    def begin(self):
        if self.enabled:
            self.cur = {}
            self.mark("start")

    def end(self):
        if self.enabled and self.cur is not None:
            self.mark("end")
            self.steps.append(self.cur)
            self.cur = None

    def summary(self):
        order = ("start", "feat_fwd_end", "hot_fwd_end", "loss_end", "hot_bwd_end", "bwd_end", "end")
        steps = [st for st in self.steps if all(k in st for k in order)]
        if not steps:
            return None
        n = len(steps)
        out = {}
        for key, (a, b) in zip(self.KEYS, zip(order[:-1], order[1:])):
            out[key + "_ms"] = round(sum(st[a].elapsed_time(st[b]) for st in steps) / n, 3)
        out["feature_cnn_ms"] = round(out["feature_cnn_fwd_ms"] + out["feature_cnn_bwd_ms"], 3)
        out["hot_path_ms"] = round(out["hot_path_fwd_ms"] + out["hot_path_bwd_ms"], 3)
        out["method"] = ("HIP events on the launching stream, mean over the timed steps; the backward cut is the last "
                         "feature-gradient hook of the step (gradient of the cost-volume builder's inputs ready)")
        return out


class KernelTimer:
    """HIP-event timing of selected C-ABI launches on the stream they are issued on (torch's current stream is the
    stream every stx_* entry point is given, stereo_toolbox_amd/ops.py:_stream)."""

    def __init__(self):
        self.records = {"conv": [], "volume": []}
        self.enabled = False

    def _timed(self, orig, units, kind):
        """The op-level wrapper computes the launch's algorithmic units; the event pair is recorded by the C-ABI call
        wrapper (`_wrap_call`) right around the launch itself -- around the whole Python op the first event would also
        cover the output allocation and argument marshalling in front of the launch (a ~9 us gap on the stream)."""
        timer = self

        def wrapper(*a, **k):
            u = units(*a, **k) if timer.enabled else None
            if u is None:
                return orig(*a, **k)
            timer._pending = (kind, u)
            try:
                return orig(*a, **k)
            finally:
                timer._pending = None
        return wrapper

    def _wrap_call(self):
        from stereo_toolbox_amd import ops
        if getattr(self, "_call_wrapped", False):
            return
        self._call_wrapped, self._pending = True, None
        orig, timer = ops._call, self
        names = {"conv": "stx_conv3d_fwd", "volume": "stx_cost_volume_fwd"}

        def call(name, *args):
            pend = timer._pending
            if pend is None or names[pend[0]] != name:
                return orig(name, *args)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(name, *args)
            e1.record()
            timer.records[pend[0]].append((e0, e1, pend[1]))
            return out
        ops._call = call

    def install_conv(self):
        """3x3x3 stride-1 Conv3d with Cin = 32, Cout <= 32: the launches served by conv3d_marchw_kernel (one launch each)."""
        from stereo_toolbox_amd import ops

        def units(x, wp, Cout, ks, stride, *a, **k):
            if not (ks == 3 and stride == 1 and Cout <= 32 and x.shape[-1] == 32):
                return None
            B, D, H, W, Cin = x.shape
            return 2.0 * B * D * H * W * Cout * Cin * 27
        self._wrap_call()
        ops.conv3d_forward = self._timed(ops.conv3d_forward, units, "conv")

    def install_volume(self):
        from stereo_toolbox_amd import ops

        def units(Lg, Rg, Lc, Rc, maxdisp, num_groups, mask_left=True, scale=None):
            n = 0
            for t in (Lg, Rg, Lc, Rc, scale):
                n += 0 if t is None else t.numel() * 4
            ref = Lg if Lg is not None else Lc
            B, _, H, W = ref.shape
            G = num_groups if Lg is not None else 0
            Cc = Lc.shape[1] if Lc is not None else 0
            return float(n + B * maxdisp * H * W * (G + 2 * Cc) * 4)
        self._wrap_call()
        ops.cost_volume_forward = self._timed(ops.cost_volume_forward, units, "volume")

    def summary(self, kind):
        recs = self.records[kind]
        if not recs:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in recs)
        un = sum(u for _, _, u in recs)
        return {"launches": len(recs), "ms_total": ms, "units_total": un}


def kernel_source_sha(*relpaths):
    """sha256 over the named kernel sources (relative to stereo_toolbox_amd/csrc) + stx_common.h: what a committed PMC summary is
    stamped with (tools/gpu/r6_h.sh) and what `committed_pmc_traffic` compares against."""
    import hashlib
    h = hashlib.sha256()
    for rel in tuple(relpaths) + ("stx_common.h",):
        with open(os.path.join(ROOT, "stereo_toolbox_amd", "csrc", rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def committed_pmc_traffic(fname, sources=None):
    """HBM bytes per launch of the dominant kernel from a COMMITTED rocprofv3 PMC summary under profiles/
    (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, KB per dispatch; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be collected inside this process: this is a constant
    of the profiled session of the same kernel (named in `traffic_source`), not a measurement of this run.
    Round 6: the summary carries the sha256 of the kernel's source files (`kernel_source_sha` line); a
    summary whose stamp does not match the sources of THIS tree (the kernel changed since it was profiled) or that has no
    stamp while `sources` is given yields None -- a stale number is never reported."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        vals, stamp = {}, None
        with open(path) as f:
            for line in f:
                parts = line.split()
                if parts and parts[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                    # (the FIRST kernel of a pass: kernel_bench times the GwcNet_GC build before the PSMNet concat build)
                    vals.setdefault(parts[0], float(line.split("avg=")[1]))
                elif parts and parts[0] == "kernel_source_sha":
                    stamp = parts[1]
        if sources is not None and stamp != kernel_source_sha(*sources):
            return None
        return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0)
    except Exception:
        return None


def same_box_fill_gbs(nbytes, dev, iters=8):
    """What a pure write stream of `nbytes` reaches on THIS box, right now: torch's `zero_()` (hipMemsetAsync-class fill
    kernel: 6.4-6.7 TB/s on the boxes of the pool, tools/ubench/store_stream) on a scratch buffer, HIP events, outside the timed
    region.  The volume builders write 82-98 % of their algorithmic bytes; no kernel can finish before a fill of its output does,
    so `achieved / this` is the fraction of the box's own write ceiling next to the fraction of the 8 TB/s data-sheet figure."""
    try:
        buf = torch.empty(int(nbytes) // 4, dtype=torch.float32, device=dev)
        for _ in range(2):
            buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            buf.zero_()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / iters
        del buf
        return nbytes / (ms * 1e-3) / 1e9
    except Exception:                    # noqa: BLE001  (platforms without event timing: the launch-path tests)
        return None


def first_existing(*names):
    for n in names:
        if os.path.exists(os.path.join(ROOT, "profiles", n)):
            return n
    return names[-1]


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, maxdisp, full_cap_s=180.0):
    """Oracle (torch-op restatement of the reference path) on the host cores.  First a BOUNDED sample of the config's
    workload (1 warm-up + 1 timed run of one pair at reduced resolution); if the sample says the config's TRUE shape fits
    into `full_cap_s` seconds, one pair at the true shape is run once and `value` is that measurement
    (`extrapolated` false) -- otherwise `value` extrapolates the sample by pixel count and says so."""
    from oracle import torch_oracle as O
    from stereo_toolbox_amd import models
    from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor
    model_name, mode, Hc, Wc, _, _ = CONFIGS[cfg]
    threads = torch.get_num_threads()

    def timed(fn):
        t0 = time.time()
        fn()
        return time.time() - t0

    if mode == "volume":
        H, W = Hc // 4, Wc // 4
        L, R = synthetic_tensor((1, 32, H, W), 1), synthetic_tensor((1, 32, H, W), 2)

        def run():
            O.build_concat_volume(L, R, maxdisp // 4)
        run()
        times = [timed(run) for _ in range(2)]
        dt = min(times)
        return {"value": round(1.0 / dt, 5), "unit": "volumes/s", "cores": threads, "cpu_model": cpu_model_name(),
                "kind": "port", "extrapolated": False, "sample_s": [round(t, 3) for t in times],
                "sample": f"oracle build_concat_volume (PSM semantics), features 1x32x{H}x{W}, D'={maxdisp // 4}; "
                          f"1 warm-up + 2 timed runs (best {dt:.3f} s)"}
    ctor = getattr(models, model_name)
    sd = ctor(maxdisp).state_dict()
    fill_state_dict(sd)
    fwd = O.acvnet_forward if model_name == "ACVNet" else (lambda s, l, r, d, **k: O.gwcnet_forward(s, l, r, d, True, **k))

    def make(H, W):
        left, right = synthetic_tensor((1, 3, H, W), 1), synthetic_tensor((1, 3, H, W), 2)
        gt = synthetic_tensor((1, H, W), 3, lo=0.0, hi=190.0)

        def run():
            if mode == "train":
                s = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
                O.smooth_l1_multi(fwd(s, left, right, maxdisp, training=True), gt, maxdisp, LOSS_W).backward()
            else:
                with torch.no_grad():
                    fwd(sd, left, right, maxdisp)
        return run
    Hs, Ws = (192, 480) if mode == "train" else (Hc // 2, Wc // 2)
    small = make(Hs, Ws)
    small()                                            # warm-up (thread pool, oneDNN primitive caches)
    ts = timed(small)
    frac = (Hs * Ws) / float(Hc * Wc)
    what = f"oracle {model_name} {'fwd+bwd' if mode == 'train' else 'eval fwd'}, 1 pair, D={maxdisp}"
    base = {"unit": "pairs/s", "cores": threads, "cpu_model": cpu_model_name(), "kind": "port"}
    if ts / frac <= full_cap_s:
        tfs = [timed(make(Hc, Wc))]                   # the config's true shape ...
        if tfs[0] <= 60.0:
            tfs.append(timed(make(Hc, Wc)))            # ... twice where the default bench run stays within a few minutes
        tf = min(tfs)
        return dict(base, value=round(1.0 / tf, 5), extrapolated=False, sample_s=[round(t, 2) for t in tfs],
                    sample=f"{what} at the config's true shape {Hc}x{Wc}: {len(tfs)} timed run(s) ({', '.join(f'{t:.1f} s' for t in tfs)}; "
                           f"value from the fastest) after a warm-up and a timed run at {Hs}x{Ws} ({ts:.1f} s, which predicted "
                           f"{ts / frac:.0f} s by pixel count)")
    return dict(base, value=round(frac / ts, 5), extrapolated=True, sample_s=[round(ts, 2)],
                sample=f"{what} at {Hs}x{Ws}; 1 warm-up + 1 timed run ({ts:.2f} s); value = measured rate x pixel ratio "
                       f"{frac:.4f} to the {Hc}x{Wc} pair (an extrapolation: the true shape was predicted to take "
                       f"{ts / frac:.0f} s, above the {full_cap_s:.0f} s cap)")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="gwc_train")
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU (default: the config's)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--maxdisp", type=int, default=192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--buckets", type=int, default=2, help="ranges of the flat gradient buffer (overlapped all-reduce)")
    ap.add_argument("--no-overlap", action="store_true", help="one all-reduce after backward() instead")
    ap.add_argument("--fused-adam", type=int, default=1, help="torch.optim.Adam(fused=True) (default) / 0 = the multi-tensor form")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (fwd+bwd+all-reduce+Adam) in one hipGraph and replay it")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Platform:
    """Where the ranks of this benchmark run: one MI355X per rank, RCCL between them.  main() talks to the machine only
    through this object, so a caller may hand in another one (the launch-path tests do, from the test tree); nothing in
    this file or in the package knows what such a substitute is made of."""
    dist_backend = "nccl"
    script = os.path.abspath(__file__)
    event_timing = True                      # HIP events per launch inside the timed region

    def device_count(self):
        return torch.cuda.device_count() if torch.cuda.is_available() else 0

    def bind(self, local):
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP hot path)")
        if local >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: LOCAL_RANK {local} but only {torch.cuda.device_count()} device(s) visible")
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def init_process_group(self, dev):
        dist.init_process_group(self.dist_backend, device_id=dev)

    def sync(self):
        torch.cuda.synchronize()

    def tuning_db(self):
        from stereo_toolbox_amd.utils import use_tuning_db
        return use_tuning_db()               # MIOpen find results of the 2-D CNN's shapes (warm-up time only)


def self_spawn(args, platform):
    """`python bench.py --gpus N` without a torchrun environment: become the launcher."""
    have = platform.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} ROCm device(s) visible on this node; refusing to run "
                         f"{args.gpus} ranks (one process per GPU, no oversubscription)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), platform.script] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes on this host driver)
    return subprocess.call(cmd, env=env)


def pin_host_threads(local, world):
    """One rank per GPU shares the node's host cores: give every rank its own contiguous block of the cores this process
    may run on (launch threads, MIOpen's solver search, the CPU side of autograd) and cap the OpenMP / torch intra-op pool
    at that block's size -- N ranks each spawning a pool as wide as the machine oversubscribe it N-fold.  Returns what
    was done (reported in the JSON line); silently does nothing where the affinity API is not available."""
    info = {"threads": None, "cores": None}
    if world <= 1:
        return info
    try:
        avail = sorted(os.sched_getaffinity(0))
        per = max(1, len(avail) // world)
        mine = avail[local * per:(local + 1) * per] or avail
        os.sched_setaffinity(0, mine)
        info["cores"] = f"{mine[0]}-{mine[-1]}"
        n = int(os.environ.get("OMP_NUM_THREADS", "0")) or per
        n = min(n, len(mine))
        os.environ["OMP_NUM_THREADS"] = str(n)
        torch.set_num_threads(n)
        info["threads"] = n
    except (AttributeError, OSError, ValueError):
        pass
    return info


def main(argv=None, platform=None):
    args = parse_args(argv)
    platform = platform or Platform()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_spawn(args, platform))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(torch.distributed.run --nproc-per-node {args.gpus})")
    pin = pin_host_threads(local, world)              # before any thread pool starts
    dev = platform.bind(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        platform.init_process_group(dev)
    sync = platform.sync

    from stereo_toolbox_amd import models, ops
    from stereo_toolbox_amd.distributed import FlatGradSync, broadcast_parameters
    from stereo_toolbox_amd.utils import fill_state_dict
    tuning_db = platform.tuning_db()

    model_name, mode, H, W, B, metric = CONFIGS[args.config]
    H, W, B = args.height or H, args.width or W, args.batch or B
    D = args.maxdisp
    torch.backends.cudnn.benchmark = True
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)                     # every rank draws its own shard (trainer_torchrun.py:88)
    timer = KernelTimer()

    split = None
    if mode == "volume":
        L = torch.randn(B, 32, H // 4, W // 4, device=dev, generator=g)
        R = torch.randn(B, 32, H // 4, W // 4, device=dev, generator=g)
        timer.install_volume()

        def step():
            return ops.cost_volume(None, None, L, R, D // 4, 0, mask_left=True)
    else:
        model = getattr(models, model_name)(D)
        sd = model.state_dict()
        fill_state_dict(sd)
        model.load_state_dict(sd)
        model = model.to(dev)
        broadcast_parameters(model)
        left = torch.randn(B, 3, H, W, device=dev, generator=g)
        right = torch.randn(B, 3, H, W, device=dev, generator=g)
        timer.install_conv()
        timer.install_volume()
        if mode == "train":
            model.train()
            overlap = not args.no_overlap and not args.graph
            gsync = FlatGradSync(model, buckets=args.buckets if overlap else 1, overlap=overlap)
            # the reference's optimizer (tests/train_torchrun.py:57: optim.Adam(model.parameters(), lr)), torch's single-kernel
            # implementation of it (fused=True: same update rule, one launch instead of ~12 multi-tensor launches)
            want_fused = bool(args.fused_adam) and dev.type == "cuda"
            try:
                opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=args.graph, fused=want_fused)
                opt_name = "torch.optim.Adam(fused=True)" if want_fused else "torch.optim.Adam"
            except (RuntimeError, TypeError):
                opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=args.graph)
                opt_name = "torch.optim.Adam (fused=True was refused by this build)"
            gt = 190.0 * torch.rand(B, H, W, device=dev, generator=g)

            sync_events = []                       # (before, after) finish(): what the gradient exchange adds to the stream
            split = PathSplit(model) if hasattr(model, "aggregate") and platform.event_timing and not args.graph else None

            def step():
                gsync.detach_grads()
                if split:
                    split.begin()
                preds = model(left, right)
                loss = smooth_l1_multi(preds, gt, D)
                if split:
                    split.mark("loss_end")
                loss.backward()
                if split:
                    split.mark("bwd_end")
                if timer.enabled and world > 1:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gsync.finish()
                    e1.record()
                    sync_events.append((e0, e1))
                else:
                    gsync.finish()
                opt.step()
                if split:
                    split.end()
        else:
            model.eval()
            split = None

            def step():
                with torch.no_grad():
                    return model(left, right)

    if args.graph and mode in ("train", "eval"):
        # hipGraph path: warm up on a side stream (MIOpen search, workspaces), then capture one step (train: forward + backward +
        # optimizer; eval: the inference forward -- ~110 launches per pair, the serving form of BASELINE config 5).
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(2, args.warmup)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        if mode == "train":
            gsync.detach_grads()
        with torch.cuda.graph(graph):
            step()
        step = graph.replay            # noqa: F811
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    sync()
    if mode == "train" and split:
        split.enabled = True
    timer.enabled = platform.event_timing and not args.graph   # per-launch HIP events cannot be recorded inside a replayed graph
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    per_rank = None
    if world > 1:
        # every rank's own wall time and the time its stream spent in the gradient exchange after backward() (pack + what
        # of the all-reduce did not hide behind the backward pass), gathered for the report; `value` uses the MAX
        exposed = 0.0
        if mode == "train" and platform.event_timing and sync_events:
            exposed = sum(a.elapsed_time(b) for a, b in sync_events) / len(sync_events)
        mine = torch.tensor([dt / args.steps * 1e3, exposed], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [round(float(x[0]), 3) for x in allr],
                    "grad_sync_ms_on_stream": [round(float(x[1]), 3) for x in allr]}
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        std_shape = (H, W, B, D) == (576, 960, 1, 192)    # the shape the committed PMC summaries were collected at

        def roof_volume():
            ks = timer.summary("volume")
            if not ks:
                return None
            ach = ks["units_total"] / (ks["ms_total"] * 1e-3) / 1e9
            src = first_existing("r06_pmc_cost_volume_fwd.txt", "r05_pmc_cost_volume_fwd.txt")
            kind = "PSMNet concat volume, fp32 copy / shift / mask" if mode == "volume" else \
                   "fused group-wise correlation + concat volume, NDHWC"
            # the same box's pure write stream of the VOLUME bytes (its output alone): the floor any builder has on this box
            vol_bytes = B * (D // 4) * (H // 4) * (W // 4) * 64 * 4
            one_kind = ks["launches"] == max(1, args.steps) or mode != "train"      # (ACVNet builds two different volumes per step)
            fill = same_box_fill_gbs(vol_bytes, dev) if (platform.event_timing and one_kind) else None
            floor_ms = vol_bytes / (fill * 1e9) * 1e3 if fill else None
            return {"bound": "hbm", "kernel": f"cost_volume_fwd_mfma_kernel ({kind})",
                    "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                    "same_box_write_stream_gbs": round(fill, 1) if fill else None,
                    "same_box_output_fill_ms": round(floor_ms, 4) if floor_ms else None,
                    "frac_of_same_box_output_fill": round(floor_ms / (ks["ms_total"] / ks["launches"]), 4) if floor_ms else None,
                    "same_box_note": "torch zero_() of a buffer of the volume's size on this box in this process (outside the "
                                     "timed region): the time a pure fill of the builder's OUTPUT takes here; boxes of the pool "
                                     "differ (5.0-6.7 TB/s for the builder's 4-KiB-row pattern, profiles/r05_store_stream_callC.txt)",
                    "traffic": committed_pmc_traffic(src, ("cost_volume_mfma.hip",) if src.startswith("r06") else None)
                    if std_shape and mode != "volume" else None,
                    "traffic_source": f"profiles/{src}: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE of the GwcNet_GC "
                                      "576x960 build, separate passes; a committed-profile constant stamped with the sha256 of the "
                                      "kernel's source (null when the source changed since, and for other shapes), not measured "
                                      "by this run",
                    "algorithmic_bytes_per_launch": int(ks["units_total"] / ks["launches"]),
                    "launches_per_step": ks["launches"] // max(1, args.steps),
                    "avg_launch_ms": round(ks["ms_total"] / ks["launches"], 4)}

        def roof_conv():
            ks = timer.summary("conv")
            if not ks:
                return None
            ach = ks["units_total"] / (ks["ms_total"] * 1e-3) / 1e12
            src = first_existing("r06_pmc_conv3d_marchw.txt", "r05_pmc_conv3d_marchw.txt")
            return {"bound": "mfma", "kernel": "conv3d_marchw_kernel (3x3x3 stride-1 Conv3d 32->32 fwd/dgrad, fp32 MFMA, weights "
                                               "resident in LDS)",
                    "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                    "traffic": committed_pmc_traffic(src, ("conv3d.hip",) if src.startswith("r06") else None) if std_shape else None,
                    "traffic_source": f"profiles/{src}: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, "
                                      "576x960 B=1 launch (algorithmic: 424 MB); a committed-profile constant stamped with the "
                                      "sha256 of the kernel's source (null when the source changed since, and for other shapes), "
                                      "not measured by this run",
                    "flop_per_launch": int(ks["units_total"] / ks["launches"]),
                    "launches_per_step": ks["launches"] // max(1, args.steps),
                    "avg_launch_ms": round(ks["ms_total"] / ks["launches"], 4)}
        # dominant kernel per config: the Conv3d march kernel for the train steps (MFMA-bound); the volume builder for the
        # volume-only config and for cfg5, for which BASELINE.json asks for the HBM roofline report (the march kernel of
        # that run is reported next to it)
        def roof_conv2d():
            """The 2-D convolution kernel at the extractor's 64 -> 64 layer shape of this run (both views of the per-GPU batch),
            timed AFTER the timed steps, 20 back-to-back forward launches between two HIP events: per-launch events inside the
            step would cost the headline ~0.9 ms (78 launches per step); the in-step average is in the committed kernel trace."""
            if not platform.event_timing:
                return None
            Bv, Hq, Wq = 2 * B, H // 4, W // 4
            x2 = torch.randn(Bv, Hq, Wq, 64, device=dev)
            w2 = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
            for _ in range(3):
                ops.conv2d_forward(x2, w2, False, 2, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv2d_forward(x2, w2, False, 2, True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            flop = 2.0 * Bv * Hq * Wq * 64 * 64 * 9
            ach = flop / (ms * 1e-3) / 1e12
            return {"bound": "mfma", "kernel": "conv2d_march_kernel (3x3 stride-1 Conv2d 64 -> 64 of the 2-D feature CNN, forward with the "
                                               "BatchNorm statistics epilogue, both views batched; fp32 MFMA 16x16x4, a 16-channel slice of "
                                               "weights in LDS; 78 such launches per GwcNet_GC train step incl. the 32-channel layers and the "
                                               "data gradients)",
                    "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "avg_launch_ms": round(ms, 4), "flop_per_launch": int(flop),
                    "measured": "after the timed steps, 20 back-to-back launches (warm caches); in the step: "
                                "profiles/r06_bench_kernel_trace_steady_callR.txt (54.5 us per launch)",
                    "launch_floor_frac": 0.83,
                    "launch_floor_note": "a launch of the same MFMA count per wave with no memory operation reaches 0.83 of the peak "
                                         "(38.4 us for 5.1 GFLOP: profiles/r06_mfma16_peak_callP.txt); MIOpen on this layer: 0.47-0.53",
                    "traffic": committed_pmc_traffic("r06_pmc_conv2d.txt", ("conv2d.hip",)) if std_shape else None,
                    "traffic_source": "profiles/r06_pmc_conv2d.txt: FETCH_SIZE (x2, gfx950) + WRITE_SIZE of this launch shape "
                                      "(algorithmic 35.4 MB), separate passes; a committed-profile constant stamped with the kernel's source"}
        extra = {}
        if mode == "train":
            roof = roof_conv()
            extra["roofline_volume_build"] = roof_volume()
            extra["roofline_conv2d"] = roof_conv2d()
        elif mode == "eval":
            roof = roof_volume()
            extra["roofline_conv3d"] = roof_conv()
        else:
            roof = roof_volume()
            if platform.event_timing:
                # "also report unpadded [1,32,135,240]" (540x960 before the reference's pad_to_2x)
                Hu = 135
                Lu, Ru = L[:, :, :Hu].contiguous(), R[:, :, :Hu].contiguous()
                for _ in range(3):
                    ops.cost_volume(None, None, Lu, Ru, D // 4, 0, mask_left=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n_it = 20
                e0.record()
                for _ in range(n_it):
                    ops.cost_volume(None, None, Lu, Ru, D // 4, 0, mask_left=True)
                e1.record()
                e1.synchronize()
                ms_u = e0.elapsed_time(e1) / n_it
                bytes_u = 2 * B * 32 * Hu * (W // 4) * 4 + B * (D // 4) * Hu * (W // 4) * 64 * 4      # 406 425 600 at B = 1
                extra["unpadded_135x240"] = {"features": f"{B}x32x{Hu}x{W // 4}", "algorithmic_bytes": bytes_u,
                                             "ms_per_volume": round(ms_u, 4), "achieved_gbs": round(bytes_u / ms_u / 1e6, 1),
                                             "frac": round(bytes_u / ms_u / 1e6 / PEAK_HBM_GBS, 4),
                                             "timing": f"{n_it} back-to-back builds between two HIP events (launch gaps included)"}
        work = {"train": f"{model_name}(maxdisp={D}) train step: fwd+bwd+allreduce+Adam",
                "eval": f"{model_name}(maxdisp={D}) inference (eval forward, no_grad), batch-parallel",
                "volume": f"build_concat_volume (PSMNet semantics) features {B}x32x{H // 4}x{W // 4}, D'={D // 4}, fwd only"}[mode]
        out = {
            "metric": metric,
            "value": round(world * B * args.steps / dt, 4),
            "unit": "pairs/s" if mode != "volume" else "volumes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"name": args.config,
                       "workload": f"{work}, {H}x{W} pairs (reference pad_to_2x shape), batch {B}/GPU, fp32, "
                                   "synthetic randn inputs, deterministic filler weights",
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "grad_sync": ("none" if mode != "train" else
                                     "none (single rank: flat fp32 buffer packed, no collective)" if world == 1 else
                                     f"flat fp32 buffer, {gsync.nb} overlapped all-reduce range(s)" if gsync.overlap else
                                     "flat fp32 buffer, 1 all-reduce after backward"),
                       "launch": "hipGraph replay" if args.graph else "eager",
                       **({"optimizer": opt_name} if mode == "train" else {}),
                       "miopen_user_db": (f"{tuning_db} (private copy of stereo_toolbox_amd/tuning/miopen)" if tuning_db
                                          else None)},
            "roofline": roof,
        }
        out.update(extra)
        if mode == "train" and split:
            ps = split.summary()
            if ps:
                out["hot_path_ms"], out["feature_cnn_ms"] = ps["hot_path_ms"], ps["feature_cnn_ms"]
                out["path_split"] = ps
        if per_rank is not None:
            out["per_rank"] = dict(per_rank, host_threads_per_rank=pin["threads"], host_cores_rank0=pin["cores"])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, D)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
