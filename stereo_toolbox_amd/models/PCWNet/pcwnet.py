"""PCWNet_G / PCWNet_GC with the multi-scale cost volumes and the 3-D aggregation on the gfx950
kernels.

Mirror of reference models/PCWNet/pcwnet.py: constructor signatures, forward(left, right) contract (train -> list of 6
predictions `[pred0, combine, pred1, pred2, pred3, disp_finetune]`, eval -> the refined disparity [B,H,W]) and
identical state-dict keys.  Four fused gwc+concat volumes (1/4, 1/8, 1/16, 1/32 resolution; the same builder kernels
as GwcNet), `hourglassup` (multi-scale fusion) + three hourglasses with Mish activations, classifier tails and the
trilinear(align_corners=True) + softmax + regression head run in stereo_toolbox_amd/csrc; the 2-D feature CNN and
the 2-D refinement network (warp, +-24 correlation, dilated residual blocks) stay stock PyTorch-ROCm.
Validated against the oracle on the host emulator and on the GPU (eval parity, full train step: tests/test_models.py
`test_pcwnet_gc_*`).  PCWNet_G constructs (state-dict compatible) but, as in the reference, cannot run (see forward()).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...aggregation import conv_block, convbn_block
from ..features2d import init_reference_style, run_pair
from .submodule import (BasicBlock, Mish, build_corrleation_volume, convbn, convbn_3d, make_layer,  # noqa: F401
                        warp)


def _head2d(cin, mid, cout):
    return nn.Sequential(convbn(cin, mid, 3, 1, 1, 1), Mish(), nn.Conv2d(mid, cout, kernel_size=1, padding=0, stride=1, bias=False))


class feature_extraction(nn.Module):
    """reference pcwnet.py:12-131 (2-D, stock torch): features at 1/4 (320 ch from l2|l3|l4), 1/8, 1/16, 1/32."""

    def __init__(self, concat_feature=False, concat_feature_channel=12):
        super().__init__()
        self.concat_feature = concat_feature
        self.firstconv = nn.Sequential(convbn(3, 32, 3, 2, 1, 1), Mish(), convbn(32, 32, 3, 1, 1, 1), Mish(),
                                       convbn(32, 32, 3, 1, 1, 1), Mish())
        p = 32
        self.layer1, p = make_layer(p, 32, 3, 1, 1, 1)
        self.layer2, p = make_layer(p, 64, 16, 2, 1, 1)
        self.layer3, p = make_layer(p, 128, 3, 1, 1, 1)
        self.layer4, p = make_layer(p, 128, 3, 1, 1, 2)
        self.layer5, p = make_layer(p, 192, 3, 2, 1, 1)
        self.layer7, p = make_layer(p, 256, 3, 2, 1, 1)
        self.layer9, p = make_layer(p, 512, 3, 2, 1, 1)
        self.gw2 = _head2d(192, 320, 320)
        self.gw3 = _head2d(256, 320, 320)
        self.gw4 = _head2d(512, 320, 320)
        self.layer11 = _head2d(320, 320, 320)
        self.layer_refine = nn.Sequential(convbn(320, 128, 3, 1, 1, 1), Mish(), convbn(128, 32, 1, 1, 0, 1), Mish())
        if concat_feature:
            self.lastconv = _head2d(320, 128, concat_feature_channel)
            self.concat2 = _head2d(192, 128, concat_feature_channel)
            self.concat3 = _head2d(256, 128, concat_feature_channel)
            self.concat4 = _head2d(512, 128, concat_feature_channel)

    def forward(self, x):
        x = self.layer1(self.firstconv(x))
        l2 = self.layer2(x)
        l3 = self.layer3(l2)
        l4 = self.layer4(l3)
        l5 = self.layer5(l4)
        l6 = self.layer7(l5)
        l7 = self.layer9(l6)
        comb = torch.cat((l2, l3, l4), dim=1)
        out = {"gw1": self.layer11(comb), "gw2": self.gw2(l5), "gw3": self.gw3(l6), "gw4": self.gw4(l7)}
        if self.concat_feature:
            out.update(concat_feature1=self.lastconv(comb), finetune_feature=self.layer_refine(comb),
                       concat_feature2=self.concat2(l5), concat_feature3=self.concat3(l6), concat_feature4=self.concat4(l7))
        return out


def _convt_bn(cin, cout):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False),
                         nn.BatchNorm3d(cout))


class hourglassup(nn.Module):
    """reference pcwnet.py:133-208: encoder that absorbs the 1/8, 1/16, 1/32 volumes on its way down (plain strided
    Conv3d, channel concat, convbn+Mish), decoder with 1x1x1 skip convolutions.  NDHWC in, NDHWC out."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv3d(c, c * 2, kernel_size=3, stride=2, padding=1, bias=False)
        self.conv2 = nn.Sequential(convbn_3d(c * 2, c * 2, 3, 1, 1), Mish())
        self.conv3 = nn.Conv3d(c * 2, c * 4, kernel_size=3, stride=2, padding=1, bias=False)
        self.conv4 = nn.Sequential(convbn_3d(c * 4, c * 4, 3, 1, 1), Mish())
        self.conv5 = nn.Conv3d(c * 4, c * 4, kernel_size=3, stride=2, padding=1, bias=False)
        self.conv6 = nn.Sequential(convbn_3d(c * 4, c * 4, 3, 1, 1), Mish())
        self.conv7 = _convt_bn(c * 4, c * 4)
        self.conv8 = _convt_bn(c * 4, c * 2)
        self.conv9 = _convt_bn(c * 2, c)
        self.combine1 = nn.Sequential(convbn_3d(c * 4, c * 2, 3, 1, 1), Mish())
        self.combine2 = nn.Sequential(convbn_3d(c * 6, c * 4, 3, 1, 1), Mish())
        self.combine3 = nn.Sequential(convbn_3d(c * 6, c * 4, 3, 1, 1), Mish())
        self.redir1 = convbn_3d(c, c, kernel_size=1, stride=1, pad=0)
        self.redir2 = convbn_3d(c * 2, c * 2, kernel_size=1, stride=1, pad=0)
        self.redir3 = convbn_3d(c * 4, c * 4, kernel_size=1, stride=1, pad=0)

    def forward(self, x, feature4, feature5, feature6):
        c1 = conv_block(x, self.conv1)                                              # 1/8, no BN, no activation
        c1 = convbn_block(torch.cat((c1, feature4), -1), self.combine1[0], mish=True)
        c2 = convbn_block(c1, self.conv2[0], mish=True)
        c3 = conv_block(c2, self.conv3)                                             # 1/16
        c3 = convbn_block(torch.cat((c3, feature5), -1), self.combine2[0], mish=True)
        c4 = convbn_block(c3, self.conv4[0], mish=True)
        c5 = conv_block(c4, self.conv5)                                             # 1/32
        c5 = convbn_block(torch.cat((c5, feature6), -1), self.combine3[0], mish=True)
        c6 = convbn_block(c5, self.conv6[0], mish=True)
        c7 = convbn_block(c6, self.conv7, mish=True, second=(c4, self.redir3))
        c8 = convbn_block(c7, self.conv8, mish=True, second=(c2, self.redir2))
        return convbn_block(c8, self.conv9, mish=True, second=(x, self.redir1))


class hourglass(nn.Module):
    """reference pcwnet.py:211-252: the GwcNet hourglass with Mish."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Sequential(convbn_3d(c, c * 2, 3, 2, 1), Mish())
        self.conv2 = nn.Sequential(convbn_3d(c * 2, c * 2, 3, 1, 1), Mish())
        self.conv3 = nn.Sequential(convbn_3d(c * 2, c * 4, 3, 2, 1), Mish())
        self.conv4 = nn.Sequential(convbn_3d(c * 4, c * 4, 3, 1, 1), Mish())
        self.conv5 = _convt_bn(c * 4, c * 2)
        self.conv6 = _convt_bn(c * 2, c)
        self.redir1 = convbn_3d(c, c, kernel_size=1, stride=1, pad=0)
        self.redir2 = convbn_3d(c * 2, c * 2, kernel_size=1, stride=1, pad=0)

    def forward(self, x):
        c1 = convbn_block(x, self.conv1[0], mish=True)
        c2 = convbn_block(c1, self.conv2[0], mish=True)
        c3 = convbn_block(c2, self.conv3[0], mish=True)
        c4 = convbn_block(c3, self.conv4[0], mish=True)
        c5 = convbn_block(c4, self.conv5, mish=True, second=(c2, self.redir2))
        return convbn_block(c5, self.conv6, mish=True, second=(x, self.redir1))


class refinenet_version3(nn.Module):
    """reference pcwnet.py:254-308 (2-D, stock torch): dilated residual stack predicting a disparity residual."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv1 = nn.Sequential(convbn(in_channels, 128, 3, 1, 1, 1), Mish())
        self.conv2 = nn.Sequential(convbn(128, 128, 3, 1, 1, 1), Mish())
        self.conv3 = nn.Sequential(convbn(128, 128, 3, 1, 2, 2), Mish())
        self.conv4 = nn.Sequential(convbn(128, 128, 3, 1, 4, 4), Mish())
        p = 128
        self.conv5, p = make_layer(p, 96, 1, 1, 1, 8)
        self.conv6, p = make_layer(p, 64, 1, 1, 1, 16)
        self.conv7, p = make_layer(p, 32, 1, 1, 1, 1)
        self.conv8 = nn.Conv2d(32, 1, kernel_size=3, padding=1, stride=1, bias=False)

    def forward(self, x, disp):
        x = self.conv4(self.conv3(self.conv2(self.conv1(x))))
        return disp + self.conv8(self.conv7(self.conv6(self.conv5(x))))


def classifier(c):
    return nn.Sequential(convbn_3d(c, c, 3, 1, 1), Mish(), nn.Conv3d(c, 1, kernel_size=3, padding=1, stride=1, bias=False))


def run_classifier(seq, x):
    """convbn_3d + Mish + Conv3d(32->1) -> dense cost [B, D', H', W']."""
    return conv_block(convbn_block(x, seq[0], mish=True), seq[2]).squeeze(-1)


class PCWNet(nn.Module):
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
        self.dres0 = nn.Sequential(convbn_3d(self.num_groups + self.concat_channels * 2, 32, 3, 1, 1), Mish(),
                                   convbn_3d(32, 32, 3, 1, 1), Mish())
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), Mish(), convbn_3d(32, 32, 3, 1, 1))
        self.combine1 = hourglassup(32)
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        self.classif0 = classifier(32)
        self.classif1 = classifier(32)
        self.classif2 = classifier(32)
        self.classif3 = classifier(32)
        self.classif4 = classifier(32)
        self.refinenet3 = refinenet_version3(146)
        self.dispupsample = nn.Sequential(convbn(1, 32, 1, 1, 0, 1), Mish())
        init_reference_style(self)

    def _volume(self, fl, fr, k, D):
        return ops.cost_volume(fl[f"gw{k}"], fr[f"gw{k}"], fl.get(f"concat_feature{k}"), fr.get(f"concat_feature{k}"), D,
                               self.num_groups, mask_left=True)

    def _refine(self, fl, fr, pred3, H, W):
        """reference pcwnet.py:472-485 / 500-512 (2-D, stock torch)."""
        pred3 = pred3.unsqueeze(1)
        left = F.interpolate(fl["finetune_feature"], [H, W], mode="bilinear", align_corners=True)
        right = F.interpolate(fr["finetune_feature"], [H, W], mode="bilinear", align_corners=True)
        right_w = warp(right, pred3)
        corr = build_corrleation_volume(left, right_w, 24, 1).squeeze(1)
        x = torch.cat((left - right_w, left, self.dispupsample(pred3), pred3, corr), dim=1)
        return self.refinenet3(x, pred3).squeeze(1)

    def forward(self, left, right):
        if not self.use_concat_volume:
            # PCWNet_G constructs with the reference's parameters (state-dict compatible) but the reference cannot run it:
            # hourglassup.conv1 is built for 64 + 40 + 24 channels (pcwnet.py:399-406 concatenate gwc + concat volumes) and
            # `finetune_feature` only exists with the concat branch (:473) -- its own forward raises a channel-mismatch
            # RuntimeError (tests/golden/state_dict_keys_pcwnet.json records it).  Same contract here, said plainly.
            raise ops.StxError("PCWNet_G (use_concat_volume=False) cannot run: the reference's architecture needs the concat "
                               "branch (hourglassup expects 128 = 64 + 40 + 24 input channels, pcwnet.py:399-406; "
                               "finetune_feature, :473) and its own forward fails the same way; use PCWNet_GC")
        fl, fr = run_pair(self.feature_extraction, left, right, self.training)
        return self.aggregate(fl, fr, left.shape[2], left.shape[3])

    @ops.fp32_region
    def aggregate(self, fl, fr, H, W):
        """Everything behind the 2-D feature CNN (pcwnet.py:388-500); fp32 also under autocast (ops.fp32_region)."""
        v1 = self._volume(fl, fr, 1, self.maxdisp // 4)
        v2 = self._volume(fl, fr, 2, self.maxdisp // 8)
        v3 = self._volume(fl, fr, 3, self.maxdisp // 16)
        v4 = self._volume(fl, fr, 4, self.maxdisp // 32)
        cost0 = convbn_block(v1, self.dres0[0], mish=True)
        cost0 = convbn_block(cost0, self.dres0[2], mish=True)
        t = convbn_block(cost0, self.dres1[0], mish=True)
        cost0 = convbn_block(t, self.dres1[2], residual=cost0)
        combine = self.combine1(cost0, v2, v3, v4)
        out1 = self.dres2(combine)
        out2 = self.dres3(out1)
        out3 = self.dres4(out2)

        def head(seq, x):
            return ops.regression_head(run_classifier(seq, x), self.maxdisp, H, W, align_corners=True)

        pred3 = head(self.classif3, out3)
        fine = self._refine(fl, fr, pred3, H, W)
        if self.training:
            return [head(self.classif0, cost0), head(self.classif4, combine), head(self.classif1, out1),
                    head(self.classif2, out2), pred3, fine]
        return fine


def PCWNet_G(d=192):
    return PCWNet(d, use_concat_volume=False)


def PCWNet_GC(d=192):
    return PCWNet(d, use_concat_volume=True)
