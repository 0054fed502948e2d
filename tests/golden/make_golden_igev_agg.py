"""Golden fixtures for the IGEV-family initial-disparity aggregation (the "small hourglass").

  python tests/golden/make_golden_igev_agg.py        (build container only: needs /root/reference)

The reference's OWN classes are executed: `BasicConv`, `FeatureAtt`, `build_gwc_volume`, `disparity_regression` from
models/IGEVStereo/submodule.py (torch / numpy only) and `hourglass` from models/IGEVStereo/igev_stereo.py.  That file's
imports pull in the 2-D backbone (`extractor.py`: `import timm_0_5_4 as timm`), absent from this image and never touched by
`hourglass`; an empty module object of that name is put into sys.modules before the package is imported from where it lies
(the same device tests/golden/make_golden_igev.py uses for torchvision).  The `IGEVStereo` class itself cannot be built
(its constructor creates the timm backbone), so the four modules of igev_stereo.py:148-151 are instantiated with the
constructor arguments written there and the forward lines :206-213 are applied to them here.

Weights come from the deterministic filler; inputs are regenerated from seeds by the tests.  Stored: eval and train outputs
(geo_encoding_volume subsampled, init_disp), a train loss, named gradient slices, BatchNorm running statistics, the state-dict
key list, and unit cases of BasicConv(k4 transposed) / FeatureAtt forward + backward: tests/golden/igev_agg.npz, .json.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.modules.setdefault("timm_0_5_4", types.ModuleType("timm_0_5_4"))
sys.path.insert(0, "/root/reference/stereo_toolbox/models")

from IGEVStereo.igev_stereo import hourglass as RefHourglass  # noqa: E402
from IGEVStereo.submodule import BasicConv as RefBasicConv  # noqa: E402
from IGEVStereo.submodule import FeatureAtt as RefFeatureAtt  # noqa: E402
from IGEVStereo.submodule import build_gwc_volume as ref_gwc  # noqa: E402
from IGEVStereo.submodule import disparity_regression as ref_regress  # noqa: E402

from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor  # noqa: E402

from tests.golden.igev_agg_config import B, CLASSIFIER_GAIN, FEAT_CH, H4, MAXDISP, W4, fill  # noqa: E402,F401


class RefAggregation(nn.Module):
    """igev_stereo.py:148-151 (constructor arguments as written there) and the forward lines :206-213."""

    def __init__(self):
        super().__init__()
        self.corr_stem = RefBasicConv(8, 8, is_3d=True, kernel_size=3, stride=1, padding=1)
        self.corr_feature_att = RefFeatureAtt(8, 96)
        self.cost_agg = RefHourglass(8)
        self.classifier = nn.Conv3d(8, 1, 3, 1, 1, bias=False)

    def forward(self, match_left, match_right, features_left):
        gwc_volume = ref_gwc(match_left, match_right, MAXDISP // 4, 8)
        gwc_volume = self.corr_stem(gwc_volume)
        gwc_volume = self.corr_feature_att(gwc_volume, features_left[0])
        geo_encoding_volume = self.cost_agg(gwc_volume, features_left)
        prob = F.softmax(self.classifier(geo_encoding_volume).squeeze(1), dim=1)
        return geo_encoding_volume, ref_regress(prob, MAXDISP // 4)


def inputs(requires_grad=False):
    ml, mr = synthetic_tensor((B, 96, H4, W4), 71), synthetic_tensor((B, 96, H4, W4), 72)
    feats = [synthetic_tensor((B, c, H4 >> i, W4 >> i), 73 + i) for i, c in enumerate(FEAT_CH)]
    if requires_grad:
        ml.requires_grad_()
        mr.requires_grad_()
        for f in feats:
            f.requires_grad_()
    return ml, mr, feats


def main():
    out = {}
    m = RefAggregation()
    sd = fill(m.state_dict())
    m.load_state_dict(sd)
    keys = [[k, list(v.shape)] for k, v in sd.items()]
    # ---- eval
    m.eval()
    with torch.no_grad():
        ml, mr, feats = inputs()
        geo, disp = m(ml, mr, feats)
    out["eval_geo"] = geo[:, :, ::2, ::2, ::2].numpy()
    out["eval_disp"] = disp.numpy()
    # ---- train step: loss on both outputs, gradients of parameters and inputs, running statistics
    m.train()
    ml, mr, feats = inputs(True)
    geo, disp = m(ml, mr, feats)
    gt = synthetic_tensor(tuple(disp.shape), 80, lo=0.0, hi=float(MAXDISP // 4 - 1))
    loss = F.smooth_l1_loss(disp, gt) + 0.1 * (geo * synthetic_tensor(tuple(geo.shape), 81)).mean()
    loss.backward()
    out["train_geo"] = geo.detach()[:, :, ::2, ::2, ::2].numpy()
    out["train_disp"] = disp.detach().numpy()
    out["train_loss"] = np.float64(loss.item())
    named = dict(m.named_parameters())
    for k in ("corr_stem.conv.weight", "corr_stem.bn.weight", "corr_feature_att.feat_att.1.bias", "cost_agg.conv1.0.conv.weight",
              "cost_agg.conv3.1.conv.weight", "cost_agg.conv3_up.conv.weight", "cost_agg.conv3_up.bn.bias",
              "cost_agg.conv2_up.conv.weight", "cost_agg.conv1_up.conv.weight", "cost_agg.agg_0.0.conv.weight",
              "cost_agg.agg_1.2.bn.weight", "cost_agg.feature_att_32.feat_att.0.conv.weight",
              "cost_agg.feature_att_up_8.feat_att.1.weight", "classifier.weight"):
        g = named[k].grad
        out["grad_" + k] = (g[:4] if g.numel() > 8192 else g).numpy()        # (big tensors: their first four slices)
    out["grad_match_left"] = ml.grad[:, ::8, ::2, ::2].numpy()
    out["grad_match_right"] = mr.grad[:, ::8, ::2, ::2].numpy()
    out["grad_feat0"] = feats[0].grad[:, ::8, ::2, ::2].numpy()
    out["grad_feat3"] = feats[3].grad[:, ::16].numpy()
    tsd = m.state_dict()
    for k in ("corr_stem.bn.running_mean", "corr_stem.bn.running_var", "cost_agg.conv3_up.bn.running_var",
              "cost_agg.agg_1.0.bn.running_mean", "cost_agg.feature_att_16.feat_att.0.bn.running_var"):
        out["stat_" + k] = tsd[k].numpy()
    out["stat_num_batches"] = tsd["cost_agg.conv2.1.bn.num_batches_tracked"].numpy()
    # ---- unit cases: transposed convolution k4 s2 p1 (+BN + LeakyReLU, train) and the gate, forward + backward
    up = RefBasicConv(16, 8, deconv=True, is_3d=True, bn=True, relu=True, kernel_size=(4, 4, 4), padding=(1, 1, 1), stride=(2, 2, 2))
    usd = up.state_dict()
    fill_state_dict(usd, seed=99)
    up.load_state_dict(usd)
    up.train()
    x = synthetic_tensor((1, 16, 3, 5, 9), 90).requires_grad_()
    y = up(x)
    y.backward(synthetic_tensor(tuple(y.shape), 91))
    out["up_y"], out["up_gx"] = y.detach().numpy(), x.grad.numpy()
    out["up_gw"], out["up_gbn"] = up.conv.weight.grad[:4].numpy(), up.bn.weight.grad.numpy()
    fa = RefFeatureAtt(8, 24)
    fsd = fa.state_dict()
    fill_state_dict(fsd, seed=98)
    fa.load_state_dict(fsd)
    fa.eval()
    cv, ft = synthetic_tensor((2, 8, 5, 4, 6), 92).requires_grad_(), synthetic_tensor((2, 24, 4, 6), 93).requires_grad_()
    z = fa(cv, ft)
    z.backward(synthetic_tensor(tuple(z.shape), 94))
    out["fa_y"], out["fa_gcv"], out["fa_gfeat"] = z.detach().numpy(), cv.grad.numpy(), ft.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "igev_agg.npz"), **out)
    with open(os.path.join(HERE, "state_dict_keys_igev_agg.json"), "w") as f:
        json.dump({"IGEVCostAggregation": keys}, f)
    print("wrote igev_agg.npz", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
