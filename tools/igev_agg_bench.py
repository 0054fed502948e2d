#!/usr/bin/env python
"""Time the IGEV-family initial-disparity aggregation (models/IGEVStereo/aggregation.py: volume -> corr_stem -> FeatureAtt ->
hourglass(8) -> classifier -> softmax + regression) at the 540x960 SceneFlow shape (padded 576x960, D=192 -> a
[B, 8, 48, 144, 240] volume), inference and one train step (forward + backward, no optimizer).  MI355X only.

  python tools/igev_agg_bench.py [--iters 20] [--batch 1] -> one JSON line
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_toolbox_amd.models.IGEVStereo import IGEVCostAggregation  # noqa: E402
from stereo_toolbox_amd.utils import fill_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--maxdisp", type=int, default=192)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = IGEVCostAggregation(a.maxdisp)
    sd = m.state_dict()
    fill_state_dict(sd, seed=4321)
    m.load_state_dict(sd)
    m = m.to(dev)
    B, H4, W4 = a.batch, a.height // 4, a.width // 4
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    ml, mr = (torch.randn(B, 96, H4, W4, device=dev, generator=g) for _ in range(2))
    feats = [torch.randn(B, c, H4 >> i, W4 >> i, device=dev, generator=g) for i, c in enumerate((96, 64, 192, 160))]

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    m.eval()

    def infer():
        with torch.no_grad():
            return m(ml, mr, feats)
    t_eval = timed(infer)
    m.train()
    mlg, mrg = ml.clone().requires_grad_(), mr.clone().requires_grad_()

    def train():
        for p in m.parameters():
            p.grad = None
        geo, disp = m(mlg, mrg, feats)
        (disp.mean() + geo.mean()).backward()
    t_train = timed(train)
    print(json.dumps({"workload": f"IGEV initial-disparity aggregation, {a.height}x{a.width}, D={a.maxdisp}, batch {B}, fp32",
                      "eval_ms": round(t_eval, 3), "train_fwd_bwd_ms": round(t_train, 3),
                      "volume": [B, 8, a.maxdisp // 4, H4, W4]}))


if __name__ == "__main__":
    main()
