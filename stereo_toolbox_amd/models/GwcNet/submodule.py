"""Drop-in functional API of reference models/GwcNet/submodule.py, backed by the HIP kernels.

Same names, argument meaning and asserts as the reference; tensors must be fp32 on a ROCm device
(no CPU fallback).  Volumes are returned with the reference's logical shape [B, C, D, H, W]; their
memory is channels-last (torch.channels_last_3d strides), which is what the MFMA aggregation
consumes -- call .contiguous() for the reference's NCDHW bytes.
"""
import torch.nn as nn

from ... import ops
from ..features2d import BasicBlock, convbn  # noqa: F401  (re-exported like the reference module)


def convbn_3d(in_channels, out_channels, kernel_size, stride, pad):
    """reference submodule.py:17-20 -- parameter container; executed by aggregation.convbn_block."""
    return nn.Sequential(nn.Conv3d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=pad,
                                   bias=False),
                         nn.BatchNorm3d(out_channels))


def disparity_regression(x, maxdisp):
    """reference submodule.py:23-27: sum_d d * x[:, d] -> [B, H, W]."""
    assert len(x.shape) == 4
    return ops.softargmax(x, maxdisp, keepdim=False)


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp):
    """reference submodule.py:30-41 (left half zeroed where w < d) -> [B, 2C, D, H, W]."""
    return ops.to_ncdhw(ops.cost_volume(None, None, refimg_fea, targetimg_fea, maxdisp, 0, mask_left=True))


def groupwise_correlation(fea1, fea2, num_groups):
    """reference submodule.py:44-50 -> [B, G, H, W] (a one-disparity gwc volume)."""
    B, C, H, W = fea1.shape
    assert C % num_groups == 0
    cost = ops.cost_volume(fea1, fea2, None, None, 1, num_groups)        # [B,1,H,W,G]
    cost = cost.reshape(B, H, W, num_groups).permute(0, 3, 1, 2)
    assert cost.shape == (B, num_groups, H, W)
    return cost


def build_gwc_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """reference submodule.py:53-63 -> [B, G, D, H, W]."""
    B, C, H, W = refimg_fea.shape
    assert C % num_groups == 0
    return ops.to_ncdhw(ops.cost_volume(refimg_fea, targetimg_fea, None, None, maxdisp, num_groups))
