"""Generates the golden fixtures under tests/golden/ by running the REFERENCE's own Python files.

Run only in the build container (needs /root/reference, read-only):  python tests/golden/make_golden.py
The reference cannot travel to the GPU box; these small .npz/.json files do.  Inputs come from the
deterministic hash generator (stereo_toolbox_amd/utils.py) so that only outputs need storing; weights
come from fill_state_dict(seed) loaded into the reference modules.

Reference import trick: `stereo_toolbox.models/__init__.py` imports every family
(needs timm, opt_einsum, ...), so the family directories are imported as namespace packages instead.
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/stereo_toolbox/models")
sys.path.insert(0, "/root/reference/stereo_toolbox")

from ACVNet.acv import ACVNet as RefACV  # noqa: E402
from ACVNet.submodule import build_concat_volume as ref_acv_concat  # noqa: E402
from GwcNet.gwcnet import GwcNet_G as RefG, GwcNet_GC as RefGC  # noqa: E402
from GwcNet.gwcnet import hourglass as RefHourGwc  # noqa: E402
from GwcNet.submodule import (build_concat_volume as ref_concat, build_gwc_volume as ref_gwc,  # noqa: E402
                              convbn_3d as ref_convbn3d, disparity_regression as ref_dr,
                              groupwise_correlation as ref_gc)
from PSMNet.stackhourglass import PSMNet as RefPSM  # noqa: E402
from PSMNet.stackhourglass import hourglass as RefHourPSM  # noqa: E402
from PSMNet.submodule import disparityregression as RefDRModule  # noqa: E402
import disparity_estimators as ref_est  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from stereo_toolbox_amd.utils import fill_state_dict, state_dict_digest, synthetic_tensor  # noqa: E402

LOSS_W = (0.5, 0.5, 0.7, 1.0)


def filled(mod):
    sd = mod.state_dict()
    fill_state_dict(sd)
    mod.load_state_dict(sd)
    return mod


def npz(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrs.items()})


def main():
    torch.manual_seed(0)
    # 1. state-dict key/shape lists (drop-in boundary)
    keys = {}
    for name, ctor in (("PSMNet", lambda: RefPSM(64)), ("GwcNet_G", lambda: RefG(64)), ("GwcNet_GC", lambda: RefGC(64)),
                       ("ACVNet", lambda: RefACV(64))):
        keys[name] = [[k, list(v.shape)] for k, v in ctor().state_dict().items()]
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f)
    print("wrote state_dict_keys.json", {k: len(v) for k, v in keys.items()})

    # 2. builders (inputs regenerated from seeds by the tests)
    out = {}
    for tag, (B, C, G, H, W, D) in (("small", (2, 16, 4, 5, 11, 6)), ("gwc320", (1, 320, 40, 8, 24, 12))):
        a, b = synthetic_tensor((B, C, H, W), 11), synthetic_tensor((B, C, H, W), 12)
        out[f"{tag}_gwc"] = ref_gwc(a, b, D, G).numpy()
        out[f"{tag}_concat"] = ref_concat(a, b, D).numpy()
        out[f"{tag}_concat_acv"] = ref_acv_concat(a, b, D).numpy()
        out[f"{tag}_gcorr"] = ref_gc(a, b, G).numpy()
    npz("builders.npz", **out)

    # 3. estimators + head chain
    peaky = torch.softmax(synthetic_tensor((2, 16, 6, 10), 13) * 4, 1)
    flat = torch.softmax(synthetic_tensor((2, 16, 6, 10), 14) * 0.01, 1)
    cost = synthetic_tensor((1, 1, 4, 5, 7), 15) * 5
    up = F.interpolate(cost, [16, 20, 28], mode="trilinear")
    head = ref_dr(F.softmax(torch.squeeze(up, 1), dim=1), 16)
    npz("estimators.npz",
        dr_peaky=ref_dr(peaky, 16).numpy(), dr_flat=ref_dr(flat, 16).numpy(),
        drmod_peaky=RefDRModule(16)(peaky).numpy(),
        softargmax_peaky=ref_est.softargmax_disparity_estimator(peaky, 16).numpy(),
        argmax_peaky=ref_est.argmax_disparity_estimator(peaky, 16).numpy(),
        argmax_flat=ref_est.argmax_disparity_estimator(flat, 16).numpy(),
        head=head.numpy())

    # 4. blocks: convbn_3d s1/s2/k1 (eval+train), the hourglass classes fwd + grads
    x = synthetic_tensor((1, 32, 8, 8, 12), 16)
    blocks = {}
    for tag, args in (("s1", (32, 32, 3, 1, 1)), ("s2", (32, 64, 3, 2, 1)), ("k1", (32, 32, 1, 1, 0))):
        m = filled(ref_convbn3d(*args))
        m.eval()
        blocks[f"convbn_{tag}_eval"] = m(x).detach().numpy()
        m.train()
        blocks[f"convbn_{tag}_train"] = m(x).detach().numpy()
        blocks[f"convbn_{tag}_rm"] = m[1].running_mean.numpy().copy()
        blocks[f"convbn_{tag}_rv"] = m[1].running_var.numpy().copy()
    hg = filled(RefHourGwc(32))
    hg.eval()
    blocks["hg_gwc_eval"] = hg(x).detach().numpy()
    hg = filled(RefHourGwc(32)).train()
    xg = x.clone().requires_grad_()
    y = hg(xg)
    blocks["hg_gwc_train"] = y.detach().numpy()
    y.square().mean().backward()
    blocks["hg_gwc_train_gx"] = xg.grad.numpy()
    blocks["hg_gwc_train_gw_conv1"] = hg.conv1[0][0].weight.grad.numpy()
    blocks["hg_gwc_train_gw_conv5"] = hg.conv5[0].weight.grad.numpy()
    blocks["hg_gwc_train_gw_redir2"] = hg.redir2[0].weight.grad.numpy()
    hp = filled(RefHourPSM(32)).eval()
    o, pre, post = hp(x, None, None)
    o2, pre2, post2 = hp(x, pre, post)
    blocks["hg_psm_eval_out"] = o.detach().numpy()
    blocks["hg_psm_eval_pre"] = pre.detach().numpy()
    blocks["hg_psm_eval_post"] = post.detach().numpy()
    blocks["hg_psm_eval_out2"] = o2.detach().numpy()
    npz("blocks.npz", **blocks)

    # 5. whole models, eval (weights: fill_state_dict(seed 1234); digest stored)
    models = {}
    left, right = synthetic_tensor((1, 3, 64, 128), 1), synthetic_tensor((1, 3, 64, 128), 2)
    with torch.no_grad():
        for tag, m in (("gwc_gc", filled(RefGC(64))), ("gwc_g", filled(RefG(64)))):
            models[f"{tag}_eval"] = m.eval()(left, right).numpy()
            models[f"{tag}_digest"] = np.int64(state_dict_digest(m.state_dict()))
        for tag, kw in (("acv", {}), ("acv_attn_only", {"attn_weights_only": True})):
            m = filled(RefACV(64, **kw)).eval()
            models[f"{tag}_eval"] = m(left, right).numpy()
            models[f"{tag}_digest"] = np.int64(state_dict_digest(m.state_dict()))
        lp, rp = synthetic_tensor((1, 3, 256, 512), 1), synthetic_tensor((1, 3, 256, 512), 2)
        m = filled(RefPSM(64)).eval()
        models["psm_eval"] = m(lp, rp).numpy()          # BASELINE.json configs[0]
        models["psm_digest"] = np.int64(state_dict_digest(m.state_dict()))
    npz("models_eval.npz", **models)

    # 6. train-mode list + loss + a handful of named grads, GwcNet_GC 64x128 D=64 B=2
    m = filled(RefGC(64)).train()
    l2, r2 = synthetic_tensor((2, 3, 64, 128), 1), synthetic_tensor((2, 3, 64, 128), 2)
    gt = synthetic_tensor((2, 64, 128), 3, lo=0.0, hi=62.0)
    preds = m(l2, r2)
    mask = (gt > 0) & (gt < 63)
    loss = sum(w * F.smooth_l1_loss(p[mask], gt[mask], reduction="mean") for p, w in zip(preds, LOSS_W))
    loss.backward()
    tr = {f"pred{i}": p.detach().numpy() for i, p in enumerate(preds)}
    tr["loss"] = np.float64(loss.item())
    named = dict(m.named_parameters())
    for k in ("dres0.0.0.weight", "dres1.2.1.weight", "dres2.conv1.0.0.weight", "dres3.conv5.0.weight",
              "dres4.redir1.0.weight", "classif3.2.weight", "classif0.0.1.bias",
              "feature_extraction.lastconv.2.weight", "feature_extraction.firstconv.0.0.weight"):
        tr["grad:" + k] = named[k].grad.numpy()
    tr["rm:dres2.conv4.0.1"] = m.dres2.conv4[0][1].running_mean.numpy().copy()
    npz("gwc_gc_train.npz", **tr)


if __name__ == "__main__":
    main()
