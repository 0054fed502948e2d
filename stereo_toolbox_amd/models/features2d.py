"""2-D feature CNNs feeding the cost-volume builders.

These are ON the critical path but NOT hand-written: they stay stock
PyTorch-ROCm (MIOpen) modules.  Only their *structure* is dictated here -- by state-dict
compatibility with the reference checkpoints (reference models/GwcNet/gwcnet.py:12-65,
models/PSMNet/submodule.py:57-132, models/ACVNet/acv.py:15-54): module names, shapes and the
Sequential indices must match so `load_checkpoint_flexible` loads published weights.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def convbn(cin, cout, k, stride, pad, dilation):
    return nn.Sequential(
        nn.Conv2d(cin, cout, k, stride, dilation if dilation > 1 else pad, dilation, bias=False),
        nn.BatchNorm2d(cout))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation), nn.ReLU(inplace=True))
        self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        if self.downsample is not None:
            x = self.downsample(x)
        return y + x


class ResTrunk(nn.Module):
    """firstconv + layer1..layer4 shared by the PSMNet / GwcNet / ACVNet extractors."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        self.firstconv = nn.Sequential(convbn(3, 32, 3, 2, 1, 1), nn.ReLU(inplace=True),
                                       convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                       convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self.layer1 = self._make_layer(32, 3, 1, 1, 1)
        self.layer2 = self._make_layer(64, 16, 2, 1, 1)
        self.layer3 = self._make_layer(128, 3, 1, 1, 1)
        self.layer4 = self._make_layer(128, 3, 1, 1, 2)

    def _make_layer(self, planes, blocks, stride, pad, dilation):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, down, pad, dilation)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def trunk(self, x):
        x = self.layer1(self.firstconv(x))
        l2 = self.layer2(x)
        l3 = self.layer3(l2)
        l4 = self.layer4(l3)
        return l2, l3, l4


def init_reference_style(model):
    """He-normal convs (fan = k^n * out_channels), BN gamma=1 beta=0, Linear bias 0 -- the
    reference's init loops (models/GwcNet/gwcnet.py:155-169 and twins)."""
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            n = m.out_channels
            for k in m.kernel_size:
                n *= k
            m.weight.data.normal_(0, math.sqrt(2.0 / n))
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.bias.data.zero_()


def run_pair(extractor, left, right, training):
    """Run a 2-D extractor on both views.  In eval mode the two views share one batched pass (same
    numbers, half the launches); in train mode they stay separate calls so that BatchNorm batch
    statistics and running-stat updates match the reference exactly (gwcnet.py:172-173)."""
    if training:
        return extractor(left), extractor(right)
    both = extractor(torch.cat((left, right), 0))
    B = left.shape[0]
    if isinstance(both, dict):
        return {k: v[:B] for k, v in both.items()}, {k: v[B:] for k, v in both.items()}
    return both[:B], both[B:]
