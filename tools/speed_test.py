"""Whole-model inference speed / memory in the style of the reference's
evaluation/speed_and_memory_test.py:11-79 (B=1, fp32, one torch.randn image used for both views, CUDA-event timing,
torch.cuda.max_memory_allocated), for the drop-in models.  MI355X only.

python tools/speed_test.py [--model GwcNet_GC|GwcNet_G|PSMNet|ACVNet] [--height 480 --width 640] [--warmup 20 --iters 100]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="GwcNet_GC")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--maxdisp", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--no-find", action="store_true", help="skip MIOpen's solver search (cudnn.benchmark off)")
    a = ap.parse_args()
    from stereo_toolbox_amd import models
    from stereo_toolbox_amd.utils import fill_state_dict, use_tuning_db
    use_tuning_db()
    torch.backends.cudnn.benchmark = not a.no_find
    dev = torch.device("cuda:0")
    model = getattr(models, a.model)(a.maxdisp)
    sd = model.state_dict()
    fill_state_dict(sd)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    x = torch.randn(1, 3, a.height, a.width, device=dev)
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        for _ in range(a.warmup):
            model(x, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            model(x, x)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(json.dumps({"model": a.model, "shape": [a.height, a.width], "maxdisp": a.maxdisp, "s_per_pair": round(ms / 1e3, 5),
                      "hz": round(1e3 / ms, 2), "max_memory_MB": round(torch.cuda.max_memory_allocated() / 2 ** 20, 1),
                      "warmup": a.warmup, "iters": a.iters, "miopen_find": not a.no_find}))


if __name__ == "__main__":
    main()
