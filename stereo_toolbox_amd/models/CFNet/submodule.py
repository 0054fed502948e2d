"""Functional API of reference models/CFNet/submodule.py for the CFNet drop-in.

The dense volume builders, the 3-D convolutions, Mish on volumes and the regressions run on the HIP kernels (the
multi-scale volumes of CFNet are GwcNet volumes at 1/8, 1/16 and 1/32 resolution).  The cascade's per-pixel search
range machinery: the variance is a HIP kernel (csrc/refine2d.hip), the uniform sampler a handful of 2-D torch ops; the sampled cost volumes of
the two cascade stages (`SpatialTransformer` gather + `groupwise_correlation_4D` + `cat`) are one HIP kernel
(`ops.sampled_volume`, csrc/sampled_volume.hip, forward + backward).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..PCWNet.submodule import (BasicBlock, FMish, Mish, build_concat_volume, build_gwc_volume, convbn,  # noqa: F401
                                convbn_3d, disparity_regression, groupwise_correlation, make_layer)


class conv2DBatchNormRelu(nn.Module):
    """reference submodule.py:70-93 (the activation is Mish despite the name)."""

    def __init__(self, in_channels, n_filters, k_size, stride, padding, bias=True, dilation=1, with_bn=True):
        super().__init__()
        conv = nn.Conv2d(int(in_channels), int(n_filters), kernel_size=k_size, padding=padding, stride=stride, bias=bias,
                         dilation=dilation if dilation > 1 else 1)
        self.cbr_unit = nn.Sequential(conv, nn.BatchNorm2d(int(n_filters)), Mish()) if with_bn else nn.Sequential(conv, Mish())

    def forward(self, x):
        return self.cbr_unit(x)


class pyramidPooling(nn.Module):
    """reference submodule.py:11-68, the configuration CFNet uses: pool_sizes=None (four pooling grids from 2 to
    min(h, w)), fusion_mode='sum', model_name='icnet': x + 0.25 * sum_i upsample(path_i(avg_pool_i(x))), halved, Mish."""

    def __init__(self, in_channels, pool_sizes=None, model_name="icnet", fusion_mode="sum", with_bn=True):
        super().__init__()
        if pool_sizes is not None or fusion_mode != "sum":
            raise NotImplementedError("only the configuration CFNet instantiates (cfnet.py:31) is on the path")
        self.path_module_list = nn.ModuleList(
            [conv2DBatchNormRelu(in_channels, in_channels, 1, 1, 0, bias=not with_bn, with_bn=with_bn) for _ in range(4)])

    def forward(self, x):
        h, w = x.shape[2:]
        sizes = [(int(h / p), int(w / p)) for p in np.linspace(2, min(h, w), 4, dtype=int)][::-1]
        pp_sum = x
        for module, k in zip(self.path_module_list, sizes):
            out = module(F.avg_pool2d(x, k, stride=k, padding=0))
            pp_sum = pp_sum + 0.25 * F.interpolate(out, size=(h, w), mode="bilinear", align_corners=False)
        return FMish(pp_sum / 2.0)


def disparity_variance(x, maxdisp, disparity):
    """reference submodule.py:128-134: sum_d p_d (d - disparity)^2 -> [B,1,H,W] (csrc/refine2d.hip, one pass)."""
    assert len(x.shape) == 4
    assert x.shape[1] == maxdisp, "the reference broadcasts arange(maxdisp) against x: the D axes must agree"
    return ops.disparity_variance(x, disparity)


def disparity_variance_confidence(x, disparity_samples, disparity):
    """reference submodule.py:136-140: sum_d p_d (disparity - sample_d)^2 -> [B,1,H,W]."""
    assert len(x.shape) == 4
    return ops.disparity_variance(x, disparity, disparity_samples)


def groupwise_correlation_4D(fea1, fea2, num_groups):
    """reference submodule.py:163-169 on [B,C,D,H,W] feature stacks."""
    B, C, D, H, W = fea1.shape
    assert C % num_groups == 0
    return (fea1 * fea2).view(B, num_groups, C // num_groups, D, H, W).mean(dim=2)


class UniformSampler(nn.Module):
    """reference submodule.py:282-303: `number_of_samples` disparities strictly inside (min, max), evenly spaced."""

    def forward(self, min_disparity, max_disparity, number_of_samples=10):
        mult = (max_disparity - min_disparity) / (number_of_samples + 1)
        k = torch.arange(1.0, number_of_samples + 1, 1, device=min_disparity.device).view(number_of_samples, 1, 1)
        return min_disparity + mult * k


class SpatialTransformer(nn.Module):
    """reference submodule.py:306-350: right features gathered at column w - sample (clamped index, zeroed where the
    un-clamped column leaves the image), left features broadcast over the samples -> two [B,C,S,H,W] stacks."""

    def forward(self, left_input, right_input, disparity_samples):
        B, C, H, W = left_input.shape
        S = disparity_samples.shape[1]
        cols = torch.arange(0.0, W, device=left_input.device).view(1, 1, 1, W)
        pos = cols - disparity_samples.float()                                  # [B,S,H,W]
        idx = pos.clamp(min=0, max=W - 1).long()
        right = right_input.unsqueeze(2).expand(B, C, S, H, W)
        warped = torch.gather(right, 4, idx.unsqueeze(1).expand(B, C, S, H, W))
        valid = 1 - ((pos < 0) | (pos > W - 1)).float().unsqueeze(1)
        return valid * warped, left_input.unsqueeze(2).expand(B, C, S, H, W)
