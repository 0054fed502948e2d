"""PSMNet (stacked hourglass) with the cost-volume hot path on hand-written gfx950 kernels.

Mirror of reference models/PSMNet/stackhourglass.py: PSMNet(maxdisp=192).forward(left, right) ->
[B,1,H,W] in eval, [pred1, pred2, pred3] in train; identical state-dict keys.
"""
import torch.nn as nn

from ... import ops
from ...aggregation import conv_block, convbn_block, deferred_bn_counters
from ..features2d import channels_last_weights_, init_reference_style, run_pair
from .submodule import convbn_3d, feature_extraction


class hourglass(nn.Module):
    """reference stackhourglass.py:10-50 (64 channels at both lower levels, no redir convs)."""

    def __init__(self, inplanes):
        super().__init__()
        c = inplanes
        self.conv1 = nn.Sequential(convbn_3d(c, c * 2, kernel_size=3, stride=2, pad=1), nn.ReLU(inplace=True))
        self.conv2 = convbn_3d(c * 2, c * 2, kernel_size=3, stride=1, pad=1)
        self.conv3 = nn.Sequential(convbn_3d(c * 2, c * 2, kernel_size=3, stride=2, pad=1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(c * 2, c * 2, kernel_size=3, stride=1, pad=1), nn.ReLU(inplace=True))
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(c * 2, c * 2, kernel_size=3, padding=1, output_padding=1,
                                                      stride=2, bias=False), nn.BatchNorm3d(c * 2))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(c * 2, c, kernel_size=3, padding=1, output_padding=1,
                                                      stride=2, bias=False), nn.BatchNorm3d(c))

    def forward(self, x, presqu, postsqu, skip=None):
        """NDHWC in/out.  `skip` (the caller's `+ cost0`, stackhourglass.py:126-132) is fused into the
        last block; the first returned value is therefore already `out + skip`."""
        out = convbn_block(x, self.conv1[0], relu=True)
        pre = convbn_block(out, self.conv2, relu=True, residual=postsqu)
        out = convbn_block(pre, self.conv3[0], relu=True)
        out = convbn_block(out, self.conv4[0], relu=True)
        post = convbn_block(out, self.conv5, relu=True, residual=presqu if presqu is not None else pre)
        out = convbn_block(post, self.conv6, relu=False, residual=skip)
        return out, pre, post


def _classifier():
    return nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True),
                         nn.Conv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False))


def _run_classifier(seq, x, add=None):
    h = convbn_block(x, seq[0], relu=True)
    cost = conv_block(h, seq[2], residual=None if add is None else add.unsqueeze(-1))
    return cost.squeeze(-1)


class PSMNet(nn.Module):
    def __init__(self, maxdisp=192):
        super().__init__()
        self.maxdisp = maxdisp
        self.feature_extraction = feature_extraction()
        self.dres0 = nn.Sequential(convbn_3d(64, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(32, 32, 3, 1, 1))
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        self.classif1 = _classifier()
        self.classif2 = _classifier()
        self.classif3 = _classifier()
        init_reference_style(self)
        channels_last_weights_(self.feature_extraction)

    def forward(self, left, right):
        with deferred_bn_counters():
            fl, fr = run_pair(self.feature_extraction, left, right, self.training)
            return self.aggregate(fl, fr, left.shape[2], left.shape[3])

    @ops.fp32_region
    def aggregate(self, fl, fr, H, W):
        """Hot path: 32-channel features at 1/4 resolution -> disparity at (H, W)."""
        # concat volume built inline in the reference (stackhourglass.py:111-120)
        cost = ops.cost_volume(None, None, fl, fr, self.maxdisp // 4, 0, mask_left=True)
        cost0 = convbn_block(cost, self.dres0[0], relu=True)
        cost0 = convbn_block(cost0, self.dres0[2], relu=True)
        t = convbn_block(cost0, self.dres1[0], relu=True)
        cost0 = convbn_block(t, self.dres1[2], relu=False, residual=cost0)
        out1, pre1, post1 = self.dres2(cost0, None, None, skip=cost0)
        out2, pre2, post2 = self.dres3(out1, pre1, post1, skip=cost0)
        out3, pre3, post3 = self.dres4(out2, pre1, post2, skip=cost0)
        cost1 = _run_classifier(self.classif1, out1)
        cost2 = _run_classifier(self.classif2, out2, add=cost1)
        cost3 = _run_classifier(self.classif3, out3, add=cost2)
        pred3 = ops.regression_head(cost3, self.maxdisp, H, W).unsqueeze(1)
        if self.training:
            pred1 = ops.regression_head(cost1, self.maxdisp, H, W).unsqueeze(1)
            pred2 = ops.regression_head(cost2, self.maxdisp, H, W).unsqueeze(1)
            return [pred1, pred2, pred3]
        return pred3
