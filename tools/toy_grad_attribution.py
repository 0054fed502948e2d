"""Where does a train-step gradient distance at the 128x256 fixture shape come from?  (GPU box; tests/golden/toy_*_path.npz)

Per tensor of the isolated 3-D path (GwcNet_GC.aggregate / ACVNet.aggregate on synthetic features), relative to the tensor's
max and against the REFERENCE's fp64 gradient samples of the fixture:
  product      the hand-written HIP path
  ref_cpu32    the reference's own fp32 run on the build container's CPU (stored in the fixture; ATen CPU accumulates BatchNorm
               sums in double)
  stock_gpu32  the oracle's torch-op restatement of the same path run on THIS chip in fp32 with stock PyTorch-ROCm kernels
               (MIOpen / ATen HIP)
  response     the product's own largest change under three relative feature perturbations of 1e-6 (conditioning)
Usage: python tools/toy_grad_attribution.py [gwc|acv]  ->  one JSON line per worst tensors + summary quantiles
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import torch_oracle as O  # noqa: E402
from stereo_toolbox_amd.losses import masked_smooth_l1_multi  # noqa: E402
from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor  # noqa: E402
from tests.golden.toy_train_config import BN_BETA_SHIFT, B, D, H, LOSS_W, W, feature_maps, sample  # noqa: E402
from tests.test_models import _toy_gold  # noqa: E402


def main(kind):
    from stereo_toolbox_amd import models
    ctor = models.ACVNet if kind == "acv" else models.GwcNet_GC
    gold = _toy_gold("toy_acv_path.npz" if kind == "acv" else "toy_gwc_gc_path.npz")
    shift = float(os.environ.get("STX_TOY_BETA_SHIFT", BN_BETA_SHIFT))      # 0: the chaotic default profile (needs fixtures made with it)
    m = ctor(D)
    sd = m.state_dict()
    fill_state_dict(sd, bn_beta_shift=shift)
    m.load_state_dict(sd)
    m = m.cuda().train()
    gt = synthetic_tensor((B, H, W), 3, lo=0.0, hi=float(D - 2)).cuda()

    def product(noise_seed=None):
        m.zero_grad(set_to_none=True)
        f = [t.cuda() for t in feature_maps(kind)]
        if noise_seed is not None:
            g = torch.Generator(device="cuda").manual_seed(noise_seed)
            f = [t * (1 + 1e-6 * torch.randn(t.shape, device="cuda", generator=g)) for t in f]
        f = [t.requires_grad_() for t in f]
        if kind == "acv":
            preds = m.aggregate(f[0], f[1], H, W, concat_left=f[2], concat_right=f[3])
        else:
            preds = m.aggregate({"gwc_feature": f[0], "concat_feature": f[2]}, {"gwc_feature": f[1], "concat_feature": f[3]}, H, W)
        masked_smooth_l1_multi(preds, gt, D, LOSS_W).backward()
        torch.cuda.synchronize()
        out = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        out.update({f"d_feature[{i}]": t.grad for i, t in enumerate(f)})
        return out

    def stock_gpu():
        torch.set_default_device("cuda")            # (the oracle creates its index / disparity tensors with the default device)
        s_ = {k: (v.detach().clone().cuda().requires_grad_("running" not in k) if v.is_floating_point() else v.clone().cuda())
              for k, v in sd.items()}
        f = [t.cuda().requires_grad_() for t in feature_maps(kind)]
        if kind == "acv":
            preds = O.acvnet_aggregate(O.Ctx(s_, True), f[0], f[1], D, H, W, cl=f[2], cr=f[3])
        else:
            preds = O.gwcnet_aggregate(O.Ctx(s_, True), f[0], f[1], f[2], f[3], D, H, W)
        O.smooth_l1_multi(preds, gt, D, LOSS_W).backward()
        out = {k: v.grad for k, v in s_.items() if v.is_floating_point() and v.grad is not None}
        out.update({f"d_feature[{i}]": t.grad for i, t in enumerate(f)})
        torch.set_default_device("cpu")
        return out

    base = product()
    pert = [product(s) for s in (1, 2, 3)]
    try:
        stock = stock_gpu()
    except Exception as e:                       # noqa: BLE001
        print(json.dumps({"stock_gpu_failed": f"{type(e).__name__}: {str(e)[:300]}"}))
        stock = {}
    rows = []
    for k in gold["names"]:
        want = gold["samples"][k].double()
        scale = float(gold["scale"][k]) + 1e-30
        e = lambda t: (sample(t.detach().cpu()).double() - want).abs().max().item() / scale  # noqa: E731
        resp = max((sample(p[k].cpu()).double() - sample(base[k].cpu()).double()).abs().max().item() for p in pert) / scale
        rows.append({"tensor": k, "product": e(base[k]), "ref_cpu32": float(gold["e32"][k]) / scale,
                     "stock_gpu32": e(stock[k]) if k in stock else None, "response_1e-6": resp})
    rows.sort(key=lambda r: -r["product"])
    for r in rows[:12]:
        print(json.dumps({k: (float(f"{v:.3e}") if isinstance(v, float) else v) for k, v in r.items()}))
    q = lambda xs, p: sorted(xs)[min(len(xs) - 1, int(p * len(xs)))]  # noqa: E731
    for key in ("product", "ref_cpu32", "stock_gpu32", "response_1e-6"):
        xs = [r[key] for r in rows if r[key] is not None]
        if xs:
            print(json.dumps({"summary": key, "kind": kind, "tensors": len(xs), "median": float(f"{q(xs, 0.5):.3e}"),
                              "p90": float(f"{q(xs, 0.9):.3e}"), "max": float(f"{max(xs):.3e}")}))
    if stock:
        rr = sorted(r["product"] / max(r["stock_gpu32"], 1e-30) for r in rows)
        print(json.dumps({"summary": "product / stock_gpu32", "median": float(f"{q(rr, 0.5):.3f}"), "p90": float(f"{q(rr, 0.9):.3f}"),
                          "max": float(f"{max(rr):.3f}")}))
        rr = sorted(r["product"] / max(r["ref_cpu32"], 1e-30) for r in rows)
        print(json.dumps({"summary": "product / ref_cpu32", "median": float(f"{q(rr, 0.5):.3f}"), "p90": float(f"{q(rr, 0.9):.3f}"),
                          "max": float(f"{max(rr):.3f}")}))
        rr = sorted(r["stock_gpu32"] / max(r["ref_cpu32"], 1e-30) for r in rows)
        print(json.dumps({"summary": "stock_gpu32 / ref_cpu32", "median": float(f"{q(rr, 0.5):.3f}"), "p90": float(f"{q(rr, 0.9):.3f}"),
                          "max": float(f"{max(rr):.3f}")}))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gwc")
