"""Functional API of reference models/PCWNet/submodule.py for the PCWNet drop-in.

The volume builders and the regression run on the HIP kernels (same entry points as GwcNet: the multi-scale volumes of
PCWNet are the GwcNet volumes at 1/4, 1/8, 1/16 and 1/32 resolution); Mish is `ops.mish`; the 2-D refinement helpers
(`warp`, the +-24-disparity `build_corrleation_volume`) act on small full-resolution 2-D maps and stay stock torch ops.
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
    -> [B, G, 2*maxdisp+1, H, W]; entry i+maxdisp pairs ref[w] with target[w - i] for i >= 0."""
    B, C, H, W = refimg_fea.shape
    assert C % num_groups == 0
    cpg = C // num_groups
    vol = refimg_fea.new_zeros(B, num_groups, 2 * maxdisp + 1, H, W)
    for i in range(-maxdisp, maxdisp + 1):
        if i > 0:
            prod = refimg_fea[..., i:] * targetimg_fea[..., :-i]
            vol[:, :, i + maxdisp, :, i:] = prod.view(B, num_groups, cpg, H, W - i).mean(2)
        elif i < 0:
            # literal reference semantics (`[:, :, :, :-i]` with negative i is the FIRST |i| columns): the first |i|
            # reference columns meet the LAST |i| target columns; all other columns of these slices stay zero
            n = -i
            prod = refimg_fea[..., :n] * targetimg_fea[..., W - n:]
            vol[:, :, i + maxdisp, :, :n] = prod.view(B, num_groups, cpg, H, n).mean(2)
        else:
            vol[:, :, maxdisp] = (refimg_fea * targetimg_fea).view(B, num_groups, cpg, H, W).mean(2)
    return vol


def warp(x, disp):
    """reference submodule.py:137-176: sample x (right view features) at column w - disp (bilinear, zeros outside,
    grid_sample's default align_corners=False on a grid normalised with W-1 / H-1 as the reference does), then zero
    every pixel whose sampling footprint left the image (validity mask < 0.999)."""
    B, C, H, W = x.shape
    xx = torch.arange(W, device=x.device, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
    yy = torch.arange(H, device=x.device, dtype=torch.float32).view(1, 1, H, 1).expand(B, 1, H, W)
    gx = 2.0 * (xx - disp) / max(W - 1, 1) - 1.0
    gy = 2.0 * yy / max(H - 1, 1) - 1.0
    grid = torch.cat((gx, gy), 1).permute(0, 2, 3, 1)
    out = F.grid_sample(x, grid, align_corners=False)
    mask = F.grid_sample(torch.ones_like(x), grid, align_corners=False)
    mask = (mask >= 0.999).to(x.dtype)
    return out * mask
