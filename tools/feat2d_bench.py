"""Times the stock-PyTorch 2-D feature extractor of GwcNet_GC (fwd+bwd, both views, 576x960) in NCHW and channels_last.

  python tools/feat2d_bench.py [--fmt nchw|nhwc|both] [--iters N] [--no-eval] [--no-benchmark]
Run under `rocprofv3 --kernel-trace --stats` (with a warm MIOPEN_USER_DB_PATH, otherwise the solver search floods the
trace) for the per-kernel split: convolutions / MIOpen BatchNorm / elementwise / layout transposes."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_toolbox_amd.models.GwcNet.gwcnet import feature_extraction  # noqa: E402
from stereo_toolbox_amd.models.features2d import channels_last_weights_  # noqa: E402
from stereo_toolbox_amd.utils import use_tuning_db  # noqa: E402

use_tuning_db()

ap = argparse.ArgumentParser()
ap.add_argument("--fmt", default="both")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--no-eval", action="store_true")
ap.add_argument("--no-benchmark", action="store_true")
ap.add_argument("--batched", action="store_true", help="both views in ONE batch-2 pass (timing experiment: what would batched "
                "convolutions with per-view BatchNorm statistics buy?)")
a = ap.parse_args()
torch.backends.cudnn.benchmark = not a.no_benchmark
dev = torch.device("cuda:0")
for fmt in (("nchw", "nhwc") if a.fmt == "both" else (a.fmt,)):
    m = feature_extraction(True, 12).to(dev).train()
    if os.environ.get("STX_FEAT2D_FUSED", "1") != "0":
        channels_last_weights_(m)       # (what the model constructors do)
    x = [torch.randn(1, 3, 576, 960, device=dev) for _ in range(2)]
    if fmt == "nhwc":
        m = m.to(memory_format=torch.channels_last)
        x = [t.contiguous(memory_format=torch.channels_last) for t in x]

    if a.batched:
        x = [torch.cat(x, 0).contiguous(memory_format=torch.channels_last if fmt == "nhwc" else torch.contiguous_format)]

    def step():
        outs = [m(t) for t in x]
        loss = sum(o["gwc_feature"].square().mean() + o["concat_feature"].square().mean() for o in outs)
        loss.backward()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(fmt, "batched" if a.batched else "per-view", "fused_glue=%s" % os.environ.get("STX_FEAT2D_FUSED", "1"), "fwd+bwd both views: %.2f ms" % (dt * 1e3), flush=True)
    if a.no_eval:
        continue
    with torch.no_grad():
        m.eval()
        for _ in range(3):
            [m(t) for t in x]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            [m(t) for t in x]
        torch.cuda.synchronize()
        print(fmt, "eval fwd both views: %.2f ms" % ((time.perf_counter() - t0) / a.iters * 1e3), flush=True)
