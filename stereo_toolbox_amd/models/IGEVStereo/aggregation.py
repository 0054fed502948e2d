"""The initial-disparity cost aggregation of the IGEV family on the HIP kernels:

    gwc_volume = build_gwc_volume(match_left, match_right, max_disp // 4, 8)           igev_stereo.py:206
    gwc_volume = corr_stem(gwc_volume)                                                  :207   BasicConv(8, 8) 3x3x3 + BN + LeakyReLU
    gwc_volume = corr_feature_att(gwc_volume, features_left[0])                         :208   FeatureAtt(8, 96)
    geo_encoding_volume = cost_agg(gwc_volume, features_left)                           :209   hourglass(8)
    prob = softmax(classifier(geo_encoding_volume).squeeze(1), dim=1)                   :212   Conv3d(8, 1, 3)
    init_disp = disparity_regression(prob, max_disp // 4)                               :213

Module names, constructor signatures and state-dict keys are the reference's (`BasicConv`, `FeatureAtt`:
models/IGEVStereo/submodule.py:9-38, 228-241; `hourglass`: igev_stereo.py:23-100), so the `corr_stem.* / corr_feature_att.* /
cost_agg.* / classifier.*` entries of a published IGEV checkpoint load into `IGEVCostAggregation` as they are.

What runs where.  Every 3-D convolution (3x3x3 stride 1 / 2, 1x1x1, and ConvTranspose3d(k4, s2, p1) as eight output-parity
classes of 2x2x2 taps -- ops.embed_deconv4_weight) is on the MFMA kernels of csrc/conv3d.hip with 8 / 16 / 32 / 48-channel
volumes (GEMM-K in steps of 8, output columns in blocks of 32); BatchNorm3d + LeakyReLU(0.01) are activation code 3 of the
fused BN passes (training) / of the convolution epilogue (inference); the FeatureAtt gate `sigmoid(att) * cv` is one
streaming kernel (stx_gate_fwd / _bwd); volume, softmax and regression are the kernels of the rest of the package.  The 2-D
side of FeatureAtt (two 1x1 Conv2d on the backbone features) is stock PyTorch, like every 2-D CNN here.  The 2-D backbone,
the GRU updates and the geometry encoding of those models are out of scope.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...aggregation import conv_block
from .submodule import init_disparity


class BasicConv(nn.Module):
    """reference IGEVStereo/submodule.py:9-38: conv (or transposed conv), optional BatchNorm, optional LeakyReLU(0.01).
    `forward` takes / returns NCDHW-logical tensors like the reference; `run` is the same block on dense NDHWC tensors."""

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, bn=True, relu=True, **kwargs):
        super().__init__()
        self.relu = relu
        self.use_bn = bn
        self.is_3d = is_3d
        if is_3d:
            self.conv = (nn.ConvTranspose3d if deconv else nn.Conv3d)(in_channels, out_channels, bias=False, **kwargs)
            if bn:
                self.bn = nn.BatchNorm3d(out_channels)
        else:
            self.conv = (nn.ConvTranspose2d if deconv else nn.Conv2d)(in_channels, out_channels, bias=False, **kwargs)
            if bn:
                self.bn = nn.BatchNorm2d(out_channels)

    def run(self, x):
        """x: dense [B, D, H, W, C] on a ROCm device."""
        return conv_block(x, self.conv, self.bn if self.use_bn else None, leaky=bool(self.relu))

    def forward(self, x):
        if self.is_3d:
            return ops.to_ncdhw(self.run(ops.to_ndhwc(x)))
        x = self.conv(x)
        if self.use_bn:
            x = self.bn(x)
        if self.relu:
            x = F.leaky_relu(x, 0.01)
        return x


class FeatureAtt(nn.Module):
    """reference IGEVStereo/submodule.py:228-241: cv * sigmoid(feat_att(feat)).unsqueeze(2)."""

    def __init__(self, cv_chan, feat_chan):
        super().__init__()
        self.feat_att = nn.Sequential(BasicConv(feat_chan, feat_chan // 2, kernel_size=1, stride=1, padding=0),
                                      nn.Conv2d(feat_chan // 2, cv_chan, 1))

    def run(self, cv, feat):
        """cv: dense [B, D, H, W, C]; feat: [B, feat_chan, H, W]."""
        att = self.feat_att(feat).permute(0, 2, 3, 1).contiguous()       # [B, H, W, C]
        return ops.gate(cv, att)

    def forward(self, cv, feat):
        return ops.to_ncdhw(self.run(ops.to_ndhwc(cv), feat))


class hourglass(nn.Module):
    """reference IGEVStereo/igev_stereo.py:23-100."""

    def __init__(self, in_channels):
        super().__init__()
        c = in_channels

        def c3(i, o, s):
            return BasicConv(i, o, is_3d=True, bn=True, relu=True, kernel_size=3, padding=1, stride=s, dilation=1)

        def up(i, o, bn, relu):
            return BasicConv(i, o, deconv=True, is_3d=True, bn=bn, relu=relu, kernel_size=(4, 4, 4), padding=(1, 1, 1),
                             stride=(2, 2, 2))

        def agg(i, o):
            return nn.Sequential(BasicConv(i, o, is_3d=True, kernel_size=1, padding=0, stride=1),
                                 BasicConv(o, o, is_3d=True, kernel_size=3, padding=1, stride=1),
                                 BasicConv(o, o, is_3d=True, kernel_size=3, padding=1, stride=1))
        self.conv1 = nn.Sequential(c3(c, c * 2, 2), c3(c * 2, c * 2, 1))
        self.conv2 = nn.Sequential(c3(c * 2, c * 4, 2), c3(c * 4, c * 4, 1))
        self.conv3 = nn.Sequential(c3(c * 4, c * 6, 2), c3(c * 6, c * 6, 1))
        self.conv3_up = up(c * 6, c * 4, True, True)
        self.conv2_up = up(c * 4, c * 2, True, True)
        self.conv1_up = up(c * 2, 8, False, False)
        self.agg_0 = agg(c * 8, c * 4)
        self.agg_1 = agg(c * 4, c * 2)
        self.feature_att_8 = FeatureAtt(c * 2, 64)
        self.feature_att_16 = FeatureAtt(c * 4, 192)
        self.feature_att_32 = FeatureAtt(c * 6, 160)
        self.feature_att_up_16 = FeatureAtt(c * 4, 192)
        self.feature_att_up_8 = FeatureAtt(c * 2, 64)

    @staticmethod
    def _seq(seq, x):
        for m in seq:
            x = m.run(x)
        return x

    def run(self, x, features):
        """x: dense [B, D, H, W, C]; features[1..3]: the 1/8, 1/16, 1/32 backbone features [B, 64 | 192 | 160, h, w]."""
        conv1 = self.feature_att_8.run(self._seq(self.conv1, x), features[1])
        conv2 = self.feature_att_16.run(self._seq(self.conv2, conv1), features[2])
        conv3 = self.feature_att_32.run(self._seq(self.conv3, conv2), features[3])
        conv3_up = self.conv3_up.run(conv3)
        conv2 = self._seq(self.agg_0, torch.cat((conv3_up, conv2), dim=-1))
        conv2 = self.feature_att_up_16.run(conv2, features[2])
        conv2_up = self.conv2_up.run(conv2)
        conv1 = self._seq(self.agg_1, torch.cat((conv2_up, conv1), dim=-1))
        conv1 = self.feature_att_up_8.run(conv1, features[1])
        return self.conv1_up.run(conv1)

    def forward(self, x, features):
        return ops.to_ncdhw(self.run(ops.to_ndhwc(x), features))


class IGEVCostAggregation(nn.Module):
    """The four modules of `IGEVStereo` that sit on the cost-volume path (igev_stereo.py:148-151), under the reference's
    attribute names, with the forward lines :206-213.  `max_disp` as `args.max_disp` (192)."""

    def __init__(self, max_disp=192):
        super().__init__()
        self.max_disp = max_disp
        self.corr_stem = BasicConv(8, 8, is_3d=True, kernel_size=3, stride=1, padding=1)
        self.corr_feature_att = FeatureAtt(8, 96)
        self.cost_agg = hourglass(8)
        self.classifier = nn.Conv3d(8, 1, 3, 1, 1, bias=False)

    @ops.fp32_region
    def forward(self, match_left, match_right, features_left):
        """match_*: [B, 96, H/4, W/4]; features_left: the four backbone levels of the left view ([B, 96, H/4, W/4] with the
        stem features concatenated, [B, 64, H/8, W/8], [B, 192, H/16, W/16], [B, 160, H/32, W/32]).
        Returns (geo_encoding_volume [B, 8, max_disp/4, H/4, W/4], init_disp [B, 1, H/4, W/4])."""
        D4 = self.max_disp // 4
        if D4 % 8 or match_left.shape[2] % 8 or match_left.shape[3] % 8:
            raise ops.StxError("IGEVCostAggregation: max_disp / 4 and the 1/4-resolution height / width must be multiples "
                               f"of 8 (three stride-2 levels), got D'={D4}, {tuple(match_left.shape[2:])}")
        vol = ops.cost_volume(match_left, match_right, None, None, D4, 8)                 # dense [B, D', H', W', 8]
        vol = self.corr_stem.run(vol)
        vol = self.corr_feature_att.run(vol, features_left[0])
        geo = self.cost_agg.run(vol, features_left)
        cost = conv_block(geo, self.classifier)                                            # [B, D', H', W', 1]
        return ops.to_ncdhw(geo), init_disparity(cost.squeeze(-1).unsqueeze(1), self.max_disp)
