"""Drop-in API of reference models/PSMNet/submodule.py (convbn_3d, disparityregression,
feature_extraction with SPP), hot-path pieces backed by the HIP kernels."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..features2d import BasicBlock, ResTrunk, convbn, run_head2d  # noqa: F401


def convbn_3d(in_planes, out_planes, kernel_size, stride, pad):
    """reference submodule.py:16-19 -- parameter container; executed by aggregation.convbn_block."""
    return nn.Sequential(nn.Conv3d(in_planes, out_planes, kernel_size=kernel_size, padding=pad, stride=stride,
                                   bias=False),
                         nn.BatchNorm3d(out_planes))


class disparityregression(nn.Module):
    """reference submodule.py:46-54: sum_d d * x[:, d], keepdim=True -> [B, 1, H, W]."""

    def __init__(self, maxdisp=192):
        super().__init__()
        self.maxdisp = maxdisp

    def forward(self, x):
        return ops.softargmax(x, self.maxdisp, keepdim=True)


class feature_extraction(ResTrunk):
    """reference submodule.py:57-132 (stock PyTorch-ROCm; SPP branches + lastconv -> 32 channels)."""

    def __init__(self):
        super().__init__()
        for i, k in ((1, 64), (2, 32), (3, 16), (4, 8)):
            setattr(self, f"branch{i}", nn.Sequential(nn.AvgPool2d((k, k), stride=(k, k)),
                                                     convbn(128, 32, 1, 1, 0, 1), nn.ReLU(inplace=True)))
        self.lastconv = nn.Sequential(convbn(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                      nn.Conv2d(128, 32, kernel_size=1, padding=0, stride=1, bias=False))

    def forward(self, x):
        raw, _, skip = self.trunk(x)
        size = (skip.shape[2], skip.shape[3])
        br = [F.interpolate(getattr(self, f"branch{i}")(skip), size, mode="bilinear", align_corners=False)
              for i in (1, 2, 3, 4)]
        feat = torch.cat((raw, skip, br[3], br[2], br[1], br[0]), 1)
        return run_head2d(self.lastconv, feat)
