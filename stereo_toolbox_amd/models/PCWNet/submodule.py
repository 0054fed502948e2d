"""Functional API of reference models/PCWNet/submodule.py for the PCWNet drop-in.

The volume builders and the regression run on the HIP kernels (same entry points as GwcNet: the multi-scale volumes of
PCWNet are the GwcNet volumes at 1/4, 1/8, 1/16 and 1/32 resolution); Mish is `ops.mish`; the 2-D refinement helpers
(`warp`, the +-24-disparity `build_corrleation_volume`) are HIP kernels too (csrc/refine2d.hip, round 5).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..GwcNet.submodule import (build_concat_volume, build_gwc_volume, convbn_3d,  # noqa: F401
                                disparity_regression, groupwise_correlation)


class Mish(nn.Module):
    """reference submodule.py:11-18: x * tanh(softplus(x)) (stock torch on 2-D features, HIP kernel on volumes)."""

    def forward(self, x):
        return F.mish(x)


def FMish(x):
    """reference submodule.py:178-190."""
    return F.mish(x)


def convbn(in_channels, out_channels, kernel_size, stride, pad, dilation):
    """reference submodule.py:21-24."""
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                   padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False),
                         nn.BatchNorm2d(out_channels))


class BasicBlock(nn.Module):
    """reference submodule.py:192-215: residual block with Mish after the first convolution only."""
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation), Mish())
        self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.conv2(self.conv1(x))
        if self.downsample is not None:
            x = self.downsample(x)
        return out + x


def make_layer(inplanes, planes, blocks, stride, pad, dilation):
    """`_make_layer` of the reference extractors (pcwnet.py:83-97, 281-295) -> (nn.Sequential, out_planes)."""
    down = None
    if stride != 1 or inplanes != planes:
        down = nn.Sequential(nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(planes))
    layers = [BasicBlock(inplanes, planes, stride, down, pad, dilation)]
    layers += [BasicBlock(planes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
    return nn.Sequential(*layers), planes


def build_corrleation_volume(refimg_fea, targetimg_fea, maxdisp, num_groups):
    """reference submodule.py:121-135 (sic): group-wise correlation for disparities -maxdisp..+maxdisp
    -> [B, G, 2*maxdisp+1, H, W]; entry i+maxdisp pairs ref[w] with target[w - i] for i >= 0; for i < 0 the literal reference
    semantics (`[:, :, :, :-i]` with negative i is the FIRST |i| columns): the first |i| reference columns meet the LAST |i|
    target columns, all other columns of those slices stay zero.  One HIP kernel pair (csrc/refine2d.hip), differentiable."""
    assert refimg_fea.shape[1] % num_groups == 0          # reference submodule.py:102 (groupwise_correlation)
    return ops.corr_volume(refimg_fea, targetimg_fea, maxdisp, num_groups)


def warp(x, disp):
    """reference submodule.py:137-176: sample x (right view features) at column w - disp (bilinear, zeros outside,
    grid_sample's default align_corners=False on a grid normalised with W-1 / H-1 as the reference does), then zero
    every pixel whose sampling footprint left the image (validity mask < 0.999).  HIP kernel (csrc/refine2d.hip): the
    footprint is computed once per pixel for all channels; differentiable in x and disp."""
    return ops.warp(x, disp)
