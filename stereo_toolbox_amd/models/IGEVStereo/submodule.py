"""The cost-volume entry of the IGEV family (IGEV-Stereo, and with the same call MonSter, MonSter/submodule.py:151-171;
FoundationStereo's volume is a DIFFERENT function -- per-group cosine similarity -- and lives in models/FoundationStereo) on the HIP
kernels -- drop-in functions of reference models/IGEVStereo/submodule.py plus the two lines of
`IGEVStereo.forward` that sit on the hot path (igev_stereo.py:206 and :211-212).  The 3-D regularisation between them
(`corr_stem`, `corr_feature_att`, `cost_agg` = hourglass(8), `classifier`) is in aggregation.py; the rest of those models
(feature backbones, GRU updates, geometry encoding) is outside the scope of this package.

    gwc_volume = build_gwc_volume(match_left, match_right, max_disp // 4, 8)      # 96 channels -> 8 groups of 12
    prob       = F.softmax(classifier(volume).squeeze(1), dim=1)
    init_disp  = disparity_regression(prob, max_disp // 4)                        # [B,1,H/4,W/4]

The volume runs on the MFMA builder with K = 12 channels per group (three v_mfma_f32_16x16x4_f32 steps per tile), its
backward on the generic gather kernel; softmax + regression on the estimator kernels.
"""
import torch

from ... import ops
from ..GwcNet.submodule import build_gwc_volume, groupwise_correlation  # noqa: F401  (same arithmetic, submodule.py:153-171)


def disparity_regression(x, maxdisp):
    """reference IGEVStereo/submodule.py:221-225: sum_d d * x[b,d,h,w], keepdim -> [B,1,H,W]."""
    return ops.softargmax(x, maxdisp, keepdim=True)


def init_gwc_volume(match_left, match_right, max_disp, num_groups=8):
    """igev_stereo.py:206 -> [B, 8, max_disp/4, H/4, W/4]."""
    return build_gwc_volume(match_left, match_right, max_disp // 4, num_groups)


class _SoftmaxDFn(torch.autograd.Function):
    """softmax over the disparity axis of a dense [B,D,H,W] cost on the HIP kernel, with its (elementwise) backward."""

    @staticmethod
    def forward(ctx, x):
        y = ops.softmax_over_d(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return y * (g - (g * y).sum(1, keepdim=True))


def init_disparity(cost, max_disp):
    """igev_stereo.py:211-212: cost [B,1,D',H',W'] (classifier output) -> init_disp [B,1,H',W'] at 1/4 resolution."""
    c = cost.squeeze(1).contiguous()
    prob = _SoftmaxDFn.apply(c) if (torch.is_grad_enabled() and c.requires_grad) else ops.softmax_over_d(c)
    return disparity_regression(prob, max_disp // 4)
