"""ACVNet with the cost-volume hot path on hand-written gfx950 kernels.

Mirror of reference models/ACVNet/acv.py: ACVNet(maxdisp=192, attn_weights_only=False,
freeze_attn_weights=False).forward(left, right); same return contracts and state-dict keys.
HIP: gwc volume, the depth-wise patch convolutions, every Conv3d/ConvTranspose3d (+BN/ReLU), the
attention-weighted concat volume `softmax(att) * concat` fused into the builder, the regression head.
Stock PyTorch-ROCm: the 2-D feature CNN / concatconv and the windowed attention block.
"""
import torch
import torch.nn as nn

from ... import ops
from ...aggregation import convbn_block, deferred_bn_counters, shared_input_convs
from ..features2d import ResTrunk, cat_features, channels_last_weights_, convbn, init_reference_style, run_head2d, run_pair
from ..GwcNet.gwcnet import classifier, run_classifier
from .submodule import attention_block, convbn_3d


class feature_extraction(ResTrunk):
    """reference acv.py:15-54: trunk only, 320-channel gwc feature."""
    fused_everywhere = True       # every BatchNorm2d of this extractor runs through features2d.conv_bn_act in train mode

    def forward(self, x):
        return {"gwc_feature": cat_features(self.trunk(x))}


class hourglass(nn.Module):
    """reference acv.py:56-93: GwcNet hourglass + windowed attention after conv4. NDHWC in/out."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Sequential(convbn_3d(c, c * 2, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(convbn_3d(c * 2, c * 2, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(convbn_3d(c * 2, c * 4, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(c * 4, c * 4, 3, 1, 1), nn.ReLU(inplace=True))
        self.attention_block = attention_block(channels_3d=c * 4, num_heads=16, block=(4, 4, 4))
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(c * 4, c * 2, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(c * 2))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(c * 2, c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(c))
        self.redir1 = convbn_3d(c, c, kernel_size=1, stride=1, pad=0)
        self.redir2 = convbn_3d(c * 2, c * 2, kernel_size=1, stride=1, pad=0)

    def forward(self, x):
        # (training: the two readers of `x` and of `c2` share an autograd node each -- aggregation.shared_input_convs)
        rx = shared_input_convs(x, [self.conv1[0], self.redir1]) or [None, None]
        c1 = convbn_block(x, self.conv1[0], relu=True, raw=rx[0])
        c2 = convbn_block(c1, self.conv2[0], relu=True)
        r2 = shared_input_convs(c2, [self.conv3[0], self.redir2]) or [None, None]
        c3 = convbn_block(c2, self.conv3[0], relu=True, raw=r2[0])
        c4 = convbn_block(c3, self.conv4[0], relu=True)
        c4 = self.attention_block(c4)
        c5 = convbn_block(c4, self.conv5, relu=True, second=(c2, self.redir2), second_raw=r2[1])
        return convbn_block(c5, self.conv6, relu=True, second=(x, self.redir1), second_raw=rx[1])


class ACVNet(nn.Module):
    def __init__(self, maxdisp=192, attn_weights_only=False, freeze_attn_weights=False):
        super().__init__()
        self.maxdisp = maxdisp
        self.attn_weights_only = attn_weights_only
        self.freeze_attn_weights = freeze_attn_weights
        self.num_groups = 40
        self.concat_channels = 32
        self.feature_extraction = feature_extraction()
        self.concatconv = nn.Sequential(convbn(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                        nn.Conv2d(128, self.concat_channels, kernel_size=1, padding=0, stride=1, bias=False))
        self.patch = nn.Conv3d(40, 40, kernel_size=(1, 3, 3), stride=1, dilation=1, groups=40, padding=(0, 1, 1), bias=False)
        self.patch_l1 = nn.Conv3d(8, 8, kernel_size=(1, 3, 3), stride=1, dilation=1, groups=8, padding=(0, 1, 1), bias=False)
        self.patch_l2 = nn.Conv3d(16, 16, kernel_size=(1, 3, 3), stride=1, dilation=2, groups=16, padding=(0, 2, 2), bias=False)
        self.patch_l3 = nn.Conv3d(16, 16, kernel_size=(1, 3, 3), stride=1, dilation=3, groups=16, padding=(0, 3, 3), bias=False)
        self.dres1_att_ = nn.Sequential(convbn_3d(40, 32, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(32, 32, 3, 1, 1))
        self.dres2_att_ = hourglass(32)
        self.classif_att_ = classifier(32)
        self.dres0 = nn.Sequential(convbn_3d(self.concat_channels * 2, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(32, 32, 3, 1, 1))
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.classif0 = classifier(32)
        self.classif1 = classifier(32)
        self.classif2 = classifier(32)
        init_reference_style(self)
        channels_last_weights_(self.feature_extraction)
        channels_last_weights_(self.concatconv)
        self._dil = {}

    def _dilations(self, device):
        key = str(device)
        if key not in self._dil:
            one = torch.ones(10, dtype=torch.int32, device=device)
            two = torch.tensor([1] * 2 + [2] * 4 + [3] * 4, dtype=torch.int32, device=device)
            self._dil[key] = (one, two)
        return self._dil[key]

    def _attention_branch(self, gl, gr):
        gwc = ops.cost_volume(gl, gr, None, None, self.maxdisp // 4, self.num_groups)      # [B,D,H,W,40]
        d1, d2 = self._dilations(gwc.device)
        v = ops.dwconv_hw(gwc, self.patch.weight.reshape(40, 9), d1)
        w2 = torch.cat((self.patch_l1.weight.reshape(8, 9), self.patch_l2.weight.reshape(16, 9),
                        self.patch_l3.weight.reshape(16, 9)), 0)
        pv = ops.dwconv_hw(v, w2, d2)                     # patch_l1/l2/l3 over channel slices, concatenated
        t = convbn_block(pv, self.dres1_att_[0], relu=True)
        ca = convbn_block(t, self.dres1_att_[2], relu=False)
        ca = self.dres2_att_(ca)
        return run_classifier(self.classif_att_, ca)      # [B, D', H', W']

    def forward(self, left, right):
        with deferred_bn_counters():
            H, W = left.shape[2], left.shape[3]
            if self.freeze_attn_weights:                  # acv.py:164-176: the extractor belongs to the frozen branch
                with torch.no_grad():
                    fl, fr = run_pair(self.feature_extraction, left, right, self.training)
            else:
                fl, fr = run_pair(self.feature_extraction, left, right, self.training)
            return self._aggregate(fl["gwc_feature"], fr["gwc_feature"], H, W)

    def aggregate(self, gl, gr, H, W, concat_left=None, concat_right=None):
        """Everything behind the feature extractor (reference acv.py:166-253) from the two 320-channel 1/4-resolution
        feature maps; `concat_left` / `concat_right` (optional) replace the `concatconv` outputs.  The cut the parity
        isolation tests use (the stock 2-D CNN on one side, the hand-written path on the other), like
        GwcNet.aggregate / PSMNet.aggregate."""
        with deferred_bn_counters():
            return self._aggregate(gl, gr, H, W, concat_left, concat_right)

    @ops.fp32_region
    def _aggregate(self, gl, gr, H, W, cl=None, cr=None):
        if self.freeze_attn_weights:
            with torch.no_grad():
                att = self._attention_branch(gl, gr)
        else:
            att = self._attention_branch(gl, gr)

        if not self.attn_weights_only:
            if cl is None:
                cl, cr = run_head2d(self.concatconv, gl), run_head2d(self.concatconv, gr)
            if torch.is_grad_enabled() and att.requires_grad:
                prob = torch.softmax(att, dim=1)          # softmax over D' (acv.py:196), tiny tensor
            else:
                prob = ops.softmax_over_d(att.contiguous())
            ac = ops.ac_volume(cl, cr, prob, self.maxdisp // 4)
            cost0 = convbn_block(ac, self.dres0[0], relu=True)
            cost0 = convbn_block(cost0, self.dres0[2], relu=True)
            t = convbn_block(cost0, self.dres1[0], relu=True)
            cost0 = convbn_block(t, self.dres1[2], relu=False, residual=cost0)
            out1 = self.dres2(cost0)
            out2 = self.dres3(out1)

        if self.training:
            preds = []
            if not self.freeze_attn_weights:
                preds.append(ops.regression_head(att, self.maxdisp, H, W))
            if not self.attn_weights_only:
                preds += [ops.regression_head(run_classifier(c, o), self.maxdisp, H, W)
                          for c, o in ((self.classif0, cost0), (self.classif1, out1), (self.classif2, out2))]
            return preds
        if self.attn_weights_only:
            return ops.regression_head(att, self.maxdisp, H, W)
        return ops.regression_head(run_classifier(self.classif2, out2), self.maxdisp, H, W)
