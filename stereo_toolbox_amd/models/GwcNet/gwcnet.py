"""GwcNet_G / GwcNet_GC with the cost-volume hot path on hand-written gfx950 kernels.

Mirror of reference models/GwcNet/gwcnet.py (constructor signatures, forward(left, right) contract,
train -> list of 4 predictions / eval -> one [B,H,W] tensor, identical state-dict keys).  The 2-D
feature CNN is stock PyTorch-ROCm; from the volume build to the disparity map everything runs in
stereo_toolbox_amd/csrc (fused gwc+concat volume, MFMA Conv3d hourglasses, fused regression head).
"""
import torch
import torch.nn as nn

from ... import ops
from ...aggregation import conv_block, convbn_block, deferred_bn_counters, shared_input_convs
from ..features2d import ResTrunk, cat_features, channels_last_weights_, convbn, init_reference_style, run_head2d, run_pair
from .submodule import convbn_3d


class feature_extraction(ResTrunk):
    """reference gwcnet.py:12-65."""
    fused_everywhere = True       # every BatchNorm2d of this extractor runs through features2d.conv_bn_act in train mode

    def __init__(self, concat_feature=False, concat_feature_channel=12):
        super().__init__()
        self.concat_feature = concat_feature
        if concat_feature:
            self.lastconv = nn.Sequential(convbn(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                          nn.Conv2d(128, concat_feature_channel, 1, bias=False))

    def forward(self, x):
        gwc = cat_features(self.trunk(x))
        if not self.concat_feature:
            return {"gwc_feature": gwc}
        return {"gwc_feature": gwc, "concat_feature": run_head2d(self.lastconv, gwc)}


class hourglass(nn.Module):
    """reference gwcnet.py:68-105; forward takes and returns NDHWC activations."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Sequential(convbn_3d(c, c * 2, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(convbn_3d(c * 2, c * 2, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(convbn_3d(c * 2, c * 4, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(c * 4, c * 4, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(c * 4, c * 2, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(c * 2))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(c * 2, c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(c))
        self.redir1 = convbn_3d(c, c, kernel_size=1, stride=1, pad=0)
        self.redir2 = convbn_3d(c * 2, c * 2, kernel_size=1, stride=1, pad=0)

    def forward(self, x, mid=None, also=None):
        """also: a further convbn_3d block that reads `x` (the classifier head of the hourglass's INPUT volume, gwcnet.py:192):
        in training its raw output comes out of the same autograd node as conv1 / redir1 (see shared_input_convs) and the
        call returns (out, raw) with `raw` for convbn_block's `raw=` argument (None when the shared path does not apply)."""
        rx = shared_input_convs(x, [self.conv1[0], self.redir1] + ([also] if also is not None else []))
        rx = rx or [None, None, None]
        c1 = convbn_block(x, self.conv1[0], relu=True, raw=rx[0])
        c2 = convbn_block(c1, self.conv2[0], relu=True)
        r2 = shared_input_convs(c2, [self.conv3[0], self.redir2]) or [None, None]
        c3 = convbn_block(c2, self.conv3[0], relu=True, raw=r2[0])
        c4 = convbn_block(c3, self.conv4[0], relu=True)
        if mid is not None:          # ACVNet inserts its windowed attention here
            c4 = mid(c4)
        c5 = convbn_block(c4, self.conv5, relu=True, second=(c2, self.redir2), second_raw=r2[1])
        out = convbn_block(c5, self.conv6, relu=True, second=(x, self.redir1), second_raw=rx[1])
        return (out, rx[2]) if also is not None else out


def classifier(c):
    return nn.Sequential(convbn_3d(c, c, 3, 1, 1), nn.ReLU(inplace=True),
                         nn.Conv3d(c, 1, kernel_size=3, padding=1, stride=1, bias=False))


def run_classifier(seq, x, add=None, raw=None):
    """convbn_3d + ReLU + Conv3d(32->1) -> dense cost [B, D', H', W'] (optionally + `add`).  raw: the first convolution's
    raw output when an hourglass call computed it together with its own readers of `x` (hourglass.forward `also`)."""
    h = convbn_block(x, seq[0], relu=True, raw=raw)
    cost = conv_block(h, seq[2], residual=None if add is None else add.unsqueeze(-1))
    return cost.squeeze(-1)


class GwcNet(nn.Module):
    def __init__(self, maxdisp, use_concat_volume=False):
        super().__init__()
        self.maxdisp = maxdisp
        self.use_concat_volume = use_concat_volume
        self.num_groups = 40
        if use_concat_volume:
            self.concat_channels = 12
            self.feature_extraction = feature_extraction(True, self.concat_channels)
        else:
            self.concat_channels = 0
            self.feature_extraction = feature_extraction(False)
        self.dres0 = nn.Sequential(convbn_3d(self.num_groups + self.concat_channels * 2, 32, 3, 1, 1),
                                   nn.ReLU(inplace=True), convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(32, 32, 3, 1, 1))
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        self.classif0 = classifier(32)
        self.classif1 = classifier(32)
        self.classif2 = classifier(32)
        self.classif3 = classifier(32)
        init_reference_style(self)
        channels_last_weights_(self.feature_extraction)

    def forward(self, left, right):
        with deferred_bn_counters():
            fl, fr = run_pair(self.feature_extraction, left, right, self.training)
            return self.aggregate(fl, fr, left.shape[2], left.shape[3])

    @ops.fp32_region
    def aggregate(self, fl, fr, H, W):
        """Everything behind the 2-D feature CNN (reference gwcnet.py:175-224): the hand-written part of the model.
        fl / fr: feature dicts of the two views ("gwc_feature" [B,320,H/4,W/4], optionally "concat_feature")."""
        vol = ops.cost_volume(fl["gwc_feature"], fr["gwc_feature"], fl.get("concat_feature"),
                              fr.get("concat_feature"), self.maxdisp // 4, self.num_groups, mask_left=True)
        cost0 = convbn_block(vol, self.dres0[0], relu=True)
        cost0 = convbn_block(cost0, self.dres0[2], relu=True)
        t = convbn_block(cost0, self.dres1[0], relu=True)
        cost0 = convbn_block(t, self.dres1[2], relu=False, residual=cost0)
        if self.training:
            # the heads of cost0 / out1 / out2 read the volume the next hourglass reads: their first convolution shares that
            # hourglass's autograd node, so the volume's gradient is accumulated in kernel epilogues (no `add` launches)
            out1, r0 = self.dres2(cost0, also=self.classif0[0])
            out2, r1 = self.dres3(out1, also=self.classif1[0])
            out3, r2 = self.dres4(out2, also=self.classif2[0])
            return [ops.regression_head(run_classifier(c, o, raw=r), self.maxdisp, H, W)
                    for c, o, r in ((self.classif0, cost0, r0), (self.classif1, out1, r1), (self.classif2, out2, r2),
                                    (self.classif3, out3, None))]
        out1 = self.dres2(cost0)
        out2 = self.dres3(out1)
        out3 = self.dres4(out2)
        return ops.regression_head(run_classifier(self.classif3, out3), self.maxdisp, H, W)


def GwcNet_G(d=192):
    return GwcNet(d, use_concat_volume=False)


def GwcNet_GC(d=192):
    return GwcNet(d, use_concat_volume=True)
