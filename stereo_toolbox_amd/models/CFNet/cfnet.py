"""CFNet (cascade and fused cost volume) with the dense cost volumes and the whole 3-D aggregation
on the gfx950 kernels.

Mirror of reference models/CFNet/cfnet.py: constructor signatures, forward(left, right) contract (train -> list of 9
predictions `[pred0_4, pred1_4, pred2_s4, pred0_s3, predmid_s3, pred1_s3_up, pred0_s2, predmid_s2, pred1_s2]`,
eval -> the 1/2-resolution cascade output upsampled, [B,H,W]) and identical state-dict keys.

  * fused stage: gwc + concat volumes at 1/8, 1/16, 1/32 resolution (the GwcNet builder kernels: 4 and 8 channels per
    group), dres0/1 stacks, `hourglassup` fusion, hourglass, classifier tails: HIP kernels (conv_block, Mish on volumes);
  * cascade stages (1/4 and 1/2 resolution): per-pixel search range from the previous stage's disparity variance, uniform
    samples, sampled gwc + concat + sample-value volume (65 / 33 channels: zero-padded to 72 / 40 for the implicit-GEMM
    kernels) -> 32- / 16-channel dres + two hourglasses + classifier tails on the HIP kernels; the sampled volume itself
    (gather + group mean) and the small 2-D range arithmetic are stock torch ops;
  * the 2-D feature pyramid (stock PyTorch-ROCm).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...aggregation import conv_block, convbn_block
from ..features2d import init_reference_style, run_pair
from ..PCWNet.pcwnet import _convt_bn, classifier, hourglass, run_classifier
from .submodule import (Mish, SpatialTransformer, UniformSampler, convbn, convbn_3d, disparity_variance,
                        disparity_variance_confidence, groupwise_correlation_4D, make_layer, pyramidPooling)


def _head2d(cin, mid, cout):
    return nn.Sequential(convbn(cin, mid, 3, 1, 1, 1), Mish(), nn.Conv2d(mid, cout, kernel_size=1, padding=0, stride=1, bias=False))


def _up(cin, cout):
    return nn.Sequential(nn.Upsample(scale_factor=2), convbn(cin, cout, 3, 1, 1, 1), Mish())


class feature_extraction(nn.Module):
    """reference cfnet.py:12-176 (2-D, stock torch): U-shaped pyramid, matching features gw2..gw6 (1/2 .. 1/32) and the
    concat features of the same levels."""

    def __init__(self, concat_feature=False, concat_feature_channel=12):
        super().__init__()
        self.concat_feature = concat_feature
        self.firstconv = nn.Sequential(convbn(3, 32, 3, 2, 1, 1), Mish(), convbn(32, 32, 3, 1, 1, 1), Mish(),
                                       convbn(32, 32, 3, 1, 1, 1), Mish())
        p = 32
        self.layer2, p = make_layer(p, 64, 1, 1, 1, 1)
        self.layer3, p = make_layer(p, 128, 1, 2, 1, 1)
        self.layer4, p = make_layer(p, 192, 1, 2, 1, 1)
        self.layer5, p = make_layer(p, 256, 1, 2, 1, 1)
        self.layer6, p = make_layer(p, 512, 1, 2, 1, 1)
        self.pyramid_pooling = pyramidPooling(512, None, fusion_mode="sum", model_name="icnet")
        self.upconv6 = _up(512, 256)
        self.iconv5 = nn.Sequential(convbn(512, 256, 3, 1, 1, 1), Mish())
        self.upconv5 = _up(256, 192)
        self.iconv4 = nn.Sequential(convbn(384, 192, 3, 1, 1, 1), Mish())
        self.upconv4 = _up(192, 128)
        self.iconv3 = nn.Sequential(convbn(256, 128, 3, 1, 1, 1), Mish())
        self.upconv3 = _up(128, 64)
        self.iconv2 = nn.Sequential(convbn(128, 64, 3, 1, 1, 1), Mish())
        self.gw2 = _head2d(64, 80, 80)
        self.gw3 = _head2d(128, 160, 160)
        self.gw4 = _head2d(192, 160, 160)
        self.gw5 = _head2d(256, 320, 320)
        self.gw6 = _head2d(512, 320, 320)
        if concat_feature:
            self.concat2 = _head2d(64, 32, concat_feature_channel // 2)
            self.concat3 = _head2d(128, 128, concat_feature_channel)
            self.concat4 = _head2d(192, 128, concat_feature_channel)
            self.concat5 = _head2d(256, 128, concat_feature_channel)
            self.concat6 = _head2d(512, 128, concat_feature_channel)

    def forward(self, x):
        x = self.firstconv(x)
        l2 = self.layer2(x)          # 1/2
        l3 = self.layer3(l2)         # 1/4
        l4 = self.layer4(l3)         # 1/8
        l5 = self.layer5(l4)         # 1/16
        l6 = self.pyramid_pooling(self.layer6(l5))     # 1/32
        d5 = self.iconv5(torch.cat((l5, self.upconv6(l6)), dim=1))
        d4 = self.iconv4(torch.cat((l4, self.upconv5(d5)), dim=1))
        d3 = self.iconv3(torch.cat((l3, self.upconv4(d4)), dim=1))
        d2 = self.iconv2(torch.cat((l2, self.upconv3(d3)), dim=1))
        out = {"gw2": self.gw2(d2), "gw3": self.gw3(d3), "gw4": self.gw4(d4)}
        if not self.concat_feature:
            return out
        out.update(gw5=self.gw5(d5), gw6=self.gw6(l6), concat_feature2=self.concat2(d2), concat_feature3=self.concat3(d3),
                   concat_feature4=self.concat4(d4), concat_feature5=self.concat5(d5), concat_feature6=self.concat6(l6))
        return out


class hourglassup(nn.Module):
    """reference cfnet.py:178-228: encoder that absorbs the 1/16 and 1/32 volumes on its way down.  NDHWC in / out."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv3d(c, c * 2, kernel_size=3, stride=2, padding=1, bias=False)
        self.conv2 = nn.Sequential(convbn_3d(c * 2, c * 2, 3, 1, 1), Mish())
        self.conv3 = nn.Conv3d(c * 2, c * 4, kernel_size=3, stride=2, padding=1, bias=False)
        self.conv4 = nn.Sequential(convbn_3d(c * 4, c * 4, 3, 1, 1), Mish())
        self.conv8 = _convt_bn(c * 4, c * 2)
        self.conv9 = _convt_bn(c * 2, c)
        self.combine1 = nn.Sequential(convbn_3d(c * 4, c * 2, 3, 1, 1), Mish())
        self.combine2 = nn.Sequential(convbn_3d(c * 6, c * 4, 3, 1, 1), Mish())
        self.combine3 = nn.Sequential(convbn_3d(c * 6, c * 4, 3, 1, 1), Mish())      # (unused by forward, as in the reference)
        self.redir1 = convbn_3d(c, c, kernel_size=1, stride=1, pad=0)
        self.redir2 = convbn_3d(c * 2, c * 2, kernel_size=1, stride=1, pad=0)
        self.redir3 = convbn_3d(c * 4, c * 4, kernel_size=1, stride=1, pad=0)          # (unused by forward)

    def forward(self, x, feature4, feature5):
        c1 = conv_block(x, self.conv1)                                                 # 1/16: plain strided conv
        c1 = convbn_block(torch.cat((c1, feature4), -1), self.combine1[0], mish=True)
        c2 = convbn_block(c1, self.conv2[0], mish=True)
        c3 = conv_block(c2, self.conv3)                                                # 1/32
        c3 = convbn_block(torch.cat((c3, feature5), -1), self.combine2[0], mish=True)
        c4 = convbn_block(c3, self.conv4[0], mish=True)
        c8 = convbn_block(c4, self.conv8, mish=True, second=(c2, self.redir2))
        return convbn_block(c8, self.conv9, mish=True, second=(x, self.redir1))


def _dres0(cin, c):
    return nn.Sequential(convbn_3d(cin, c, 3, 1, 1), Mish(), convbn_3d(c, c, 3, 1, 1), Mish())


def _dres1(c):
    return nn.Sequential(convbn_3d(c, c, 3, 1, 1), Mish(), convbn_3d(c, c, 3, 1, 1))


def _run_dres(x, d0, d1):
    """cost0 = dres0(x); cost0 = dres1(cost0) + cost0 (cfnet.py:529-535)."""
    c = convbn_block(x, d0[0], mish=True)
    c = convbn_block(c, d0[2], mish=True)
    t = convbn_block(c, d1[0], mish=True)
    return convbn_block(t, d1[2], residual=c)


def _up2d(x, scale, H, W):
    return F.interpolate(x * scale, [H, W], mode="bilinear", align_corners=True)


class cfnet(nn.Module):
    def __init__(self, maxdisp, use_concat_volume=False):
        super().__init__()
        self.maxdisp = maxdisp
        self.use_concat_volume = use_concat_volume
        self.v_scale_s1, self.v_scale_s2, self.v_scale_s3 = 1, 2, 3
        self.sample_count_s1, self.sample_count_s2, self.sample_count_s3 = 6, 10, 14
        self.num_groups = 40
        self.uniform_sampler = UniformSampler()
        self.spatial_transformer = SpatialTransformer()
        if use_concat_volume:
            self.concat_channels = 12
            self.feature_extraction = feature_extraction(True, self.concat_channels)
        else:
            self.concat_channels = 0
            self.feature_extraction = feature_extraction(False)
        G, Cc = self.num_groups, self.concat_channels
        self.dres0, self.dres1 = _dres0(G + Cc * 2, 32), _dres1(32)
        self.dres0_5, self.dres1_5 = _dres0(G + Cc * 2, 64), _dres1(64)
        self.dres0_6, self.dres1_6 = _dres0(G + Cc * 2, 64), _dres1(64)
        self.combine1 = hourglassup(32)
        self.dres3 = hourglass(32)
        self.confidence0_s3, self.confidence1_s3 = _dres0(G + Cc * 2 + 1, 32), _dres1(32)
        self.confidence2_s3 = hourglass(32)
        self.confidence3_s3 = hourglass(32)
        self.confidence0_s2, self.confidence1_s2 = _dres0(G // 2 + Cc + 1, 16), _dres1(16)
        self.confidence2_s2 = hourglass(16)
        self.confidence3_s2 = hourglass(16)
        self.confidence_classif0_s3 = classifier(32)
        self.confidence_classif1_s3 = classifier(32)
        self.confidence_classifmid_s3 = classifier(32)
        self.confidence_classif0_s2 = classifier(16)
        self.confidence_classif1_s2 = classifier(16)
        self.confidence_classifmid_s2 = classifier(16)
        self.classif0 = classifier(32)
        self.classif1 = classifier(32)
        self.classif2 = classifier(32)
        self.gamma_s3 = nn.Parameter(torch.zeros(1))
        self.beta_s3 = nn.Parameter(torch.zeros(1))
        self.gamma_s2 = nn.Parameter(torch.zeros(1))
        self.beta_s2 = nn.Parameter(torch.zeros(1))
        # test hook (None in normal use): (samples_s3, samples_s2) replacing the integer disparity samples of the two
        # cascade stages, so that parity tests can compare train-mode outputs with the samples the reference drew (one
        # flipped integer sample moves every cascade prediction under batch-stat BN; tests/golden/make_golden_cfnet.py)
        self.forced_samples = None
        init_reference_style(self)

    # ------------------------------------------------------------------ cascade helpers (cfnet.py:436-497)
    def generate_search_range(self, sample_count, input_min_disparity, input_max_disparity, scale):
        hi = self.maxdisp // (2 ** scale) - 1
        slack = torch.clamp(sample_count - input_max_disparity + input_min_disparity, min=0) / 2.0
        return (torch.clamp(input_min_disparity - slack, min=0, max=hi), torch.clamp(input_max_disparity + slack, min=0, max=hi))

    def generate_disparity_samples(self, min_disparity, max_disparity, sample_count=12):
        samples = self.uniform_sampler(min_disparity, max_disparity, sample_count)
        return torch.cat((torch.floor(min_disparity), samples, torch.ceil(max_disparity)), dim=1).long()

    def cost_volume_generator(self, left_input, right_input, disparity_samples, model="concat", num_groups=40):
        right_map, left_map = self.spatial_transformer(left_input, right_input, disparity_samples)
        disparity_samples = disparity_samples.unsqueeze(1).float()
        if model == "concat":
            return torch.cat((left_map, right_map), dim=1), disparity_samples
        return groupwise_correlation_4D(left_map, right_map, num_groups), disparity_samples

    def _volume(self, fl, fr, k, D):
        return ops.cost_volume(fl[f"gw{k}"], fr[f"gw{k}"], fl.get(f"concat_feature{k}"), fr.get(f"concat_feature{k}"), D,
                               self.num_groups, mask_left=True)

    def _stage(self, fl, fr, k, lo, hi, count, scale, groups, d0, d1, hg2, hg3, classif1):
        """One cascade stage (cfnet.py:557-571 / 588-602): samples -> sampled volume -> aggregation -> distribution over
        the samples.  Returns (cost0, out1, possibility, samples [B,S,H,W])."""
        lo1, hi1 = self.generate_search_range(count + 1, lo, hi, scale=scale)
        samples = self.generate_disparity_samples(lo1, hi1, count).float()
        if self.forced_samples is not None:
            samples = self.forced_samples[0 if k == 3 else 1].to(samples)
        # gather + group-wise correlation + concat + hypothesis channel in one kernel (stx_sampled_volume_fwd); the
        # torch-op restatement of the reference's chain stays available as `cost_volume_generator` (API parity)
        vol = ops.sampled_volume(fl[f"gw{k}"], fr[f"gw{k}"], fl.get(f"concat_feature{k}"), fr.get(f"concat_feature{k}"), samples, groups)
        cost0 = _run_dres(vol, d0, d1)
        out1 = hg2(cost0)
        out2 = hg3(out1)
        poss = F.softmax(run_classifier(classif1, out2), dim=1)
        return cost0, out1, poss, samples

    def forward(self, left, right):
        fl, fr = run_pair(self.feature_extraction, left, right, self.training)
        return self.aggregate(fl, fr, left.shape[2], left.shape[3])

    @ops.fp32_region
    def aggregate(self, fl, fr, H, W):
        """Everything behind the 2-D feature CNN (cfnet.py:502-640); fp32 also under autocast (ops.fp32_region)."""
        # ---- fused stage at 1/8 with the 1/16 and 1/32 volumes injected (cfnet.py:502-541)
        v4 = self._volume(fl, fr, 4, self.maxdisp // 8)
        v5 = self._volume(fl, fr, 5, self.maxdisp // 16)
        v6 = self._volume(fl, fr, 6, self.maxdisp // 32)
        cost0_4 = _run_dres(v4, self.dres0, self.dres1)
        cost0_5 = _run_dres(v5, self.dres0_5, self.dres1_5)
        cost0_6 = _run_dres(v6, self.dres0_6, self.dres1_6)
        out1_4 = self.combine1(cost0_4, cost0_5, cost0_6)
        out2_4 = self.dres3(out1_4)
        poss_s4 = F.softmax(run_classifier(self.classif2, out2_4), dim=1)
        pred2_s4 = ops.softargmax(poss_s4, self.maxdisp // 8, keepdim=True)
        cur = pred2_s4.detach()
        var = disparity_variance(poss_s4, self.maxdisp // 8, cur).sqrt()
        lo = _up2d(cur - (self.gamma_s3 + 1) * var - self.beta_s3, 2, H // 4, W // 4)
        hi = _up2d(cur + (self.gamma_s3 + 1) * var + self.beta_s3, 2, H // 4, W // 4)
        # ---- cascade stage at 1/4 (cfnet.py:553-571)
        cost0_s3, out1_s3, poss_s3, samples_s3 = self._stage(fl, fr, 3, lo, hi, self.sample_count_s3, 2, self.num_groups,
                                                            self.confidence0_s3, self.confidence1_s3, self.confidence2_s3,
                                                            self.confidence3_s3, self.confidence_classif1_s3)
        pred1_s3 = torch.sum(poss_s3 * samples_s3, dim=1, keepdim=True)
        cur = pred1_s3.detach()
        var = disparity_variance_confidence(poss_s3, samples_s3, cur).sqrt()
        lo = _up2d(cur - (self.gamma_s2 + 1) * var - self.beta_s2, 2, H // 2, W // 2)
        hi = _up2d(cur + (self.gamma_s2 + 1) * var + self.beta_s2, 2, H // 2, W // 2)
        # ---- cascade stage at 1/2 (cfnet.py:584-602)
        cost0_s2, out1_s2, poss_s2, samples_s2 = self._stage(fl, fr, 2, lo, hi, self.sample_count_s2, 1, self.num_groups // 2,
                                                            self.confidence0_s2, self.confidence1_s2, self.confidence2_s2,
                                                            self.confidence3_s2, self.confidence_classif1_s2)
        pred1_s2 = torch.sum(poss_s2 * samples_s2, dim=1, keepdim=True)
        if not self.training:
            return _up2d(pred1_s2, 2, H, W).squeeze(1)

        def head(seq, x):
            return ops.regression_head(run_classifier(seq, x), self.maxdisp, H, W, align_corners=True)

        def sampled(seq, x, samples, scale):
            p = F.softmax(run_classifier(seq, x), dim=1)
            return _up2d(torch.sum(p * samples, dim=1, keepdim=True), scale, H, W).squeeze(1)

        return [head(self.classif0, cost0_4), head(self.classif1, out1_4), _up2d(pred2_s4, 8, H, W).squeeze(1),
                sampled(self.confidence_classif0_s3, cost0_s3, samples_s3, 4),
                sampled(self.confidence_classifmid_s3, out1_s3, samples_s3, 4), _up2d(pred1_s3, 4, H, W).squeeze(1),
                sampled(self.confidence_classif0_s2, cost0_s2, samples_s2, 2),
                sampled(self.confidence_classifmid_s2, out1_s2, samples_s2, 2), _up2d(pred1_s2, 2, H, W).squeeze(1)]


def CFNet(d=192):
    return cfnet(d, use_concat_volume=True)
