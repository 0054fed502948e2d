"""ORACLE (test infrastructure -- never imported by the product path).

A CPU restatement, in stock fp32 torch ops, of the reference's cost-volume hot path.  Only
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; it is
the *checker*, never the thing measured or shipped.

Pinning: every function here is compared against the reference's own Python files (imported from
/root/reference in the build container by tests/golden/make_golden.py) and the resulting
input/output vectors are committed under tests/golden/*.npz; tests/test_oracle_golden.py replays
them.  The reference ships no tests/golden vectors of its own for this path.

Everything is *functional over a state-dict*: `sd` maps the reference's parameter names
(`dres0.0.0.weight`, `dres2.conv5.1.running_var`, ...) to tensors, so the same dict can be loaded
into the reference modules, into the product modules and evaluated here.

Citations are relative to /root/reference/stereo_toolbox/models.
"""
import math

import torch
import torch.nn.functional as F

EPS = 1e-5
MOMENTUM = 0.1


# ----------------------------------------------------------------------------- volume builders
def groupwise_correlation(fea1, fea2, num_groups):
    """GwcNet/submodule.py:44-50: per-group mean of channel products."""
    B, C, H, W = fea1.shape
    assert C % num_groups == 0
    return (fea1 * fea2).view(B, num_groups, C // num_groups, H, W).mean(dim=2)


def _shift_right(t, maxdisp):
    """[B,C,H,W] -> [B,C,D,H,W] with out[..., d, h, w] = w>=d ? t[..., h, w-d] : 0 (gather form)."""
    B, C, H, W = t.shape
    w = torch.arange(W).view(1, W)
    d = torch.arange(maxdisp).view(maxdisp, 1)
    idx = (w - d).clamp(min=0)                     # [D, W]
    valid = (w >= d).to(t.dtype)                   # [D, W]
    g = t[:, :, :, idx]                            # [B,C,H,D,W]
    g = g * valid.view(1, 1, 1, maxdisp, W)
    return g.permute(0, 1, 3, 2, 4).contiguous(), valid


def build_gwc_volume(ref, tgt, maxdisp, num_groups):
    """GwcNet/submodule.py:53-63 (dup ACVNet/submodule.py:228-238)."""
    B, C, H, W = ref.shape
    assert C % num_groups == 0
    cpg = C // num_groups
    shifted, _ = _shift_right(tgt, maxdisp)                       # [B,C,D,H,W]
    prod = ref.unsqueeze(2) * shifted
    return prod.view(B, num_groups, cpg, maxdisp, H, W).mean(dim=2)


def build_concat_volume(ref, tgt, maxdisp, mask_left=True):
    """mask_left=True: GwcNet/submodule.py:30-41 and PSMNet/stackhourglass.py:111-120.
    mask_left=False: ACVNet/submodule.py:180-191 (left feature copied to every column)."""
    B, C, H, W = ref.shape
    right, valid = _shift_right(tgt, maxdisp)
    left = ref.unsqueeze(2).expand(B, C, maxdisp, H, W)
    if mask_left:
        left = left * valid.view(1, 1, maxdisp, 1, W)
    return torch.cat((left, right), dim=1).contiguous()


def fs_groupwise_correlation(fea1, fea2, num_groups):
    """FoundationStereo/submodule.py:388-397: per-group cosine similarity -- both maps L2-normalised over each group's
    channels (F.normalize(dim=2), eps 1e-12, fp32), products SUMMED over the group."""
    B, C, H, W = fea1.shape
    assert C % num_groups == 0
    cpg = C // num_groups
    a = F.normalize(fea1.reshape(B, num_groups, cpg, H, W).float(), dim=2)
    b = F.normalize(fea2.reshape(B, num_groups, cpg, H, W).float(), dim=2)
    return (a * b).sum(dim=2)


def fs_build_gwc_volume(ref, tgt, maxdisp, num_groups):
    """FoundationStereo/submodule.py:399-413.  The norm of a pixel's group does not depend on the disparity: normalise both
    maps once, shift the right one (gather form, as build_gwc_volume above), sum the products."""
    B, C, H, W = ref.shape
    assert C % num_groups == 0
    cpg = C // num_groups
    a = F.normalize(ref.reshape(B, num_groups, cpg, H, W).float(), dim=2).reshape(B, C, H, W)
    b = F.normalize(tgt.reshape(B, num_groups, cpg, H, W).float(), dim=2).reshape(B, C, H, W)
    shifted, _ = _shift_right(b, maxdisp)
    return (a.unsqueeze(2) * shifted).view(B, num_groups, cpg, maxdisp, H, W).sum(dim=2)


def fs_build_concat_volume(ref, tgt, maxdisp):
    """FoundationStereo/submodule.py:416-427: the left half is the left map at every disparity (not masked)."""
    return build_concat_volume(ref, tgt, maxdisp, mask_left=False)


# ----------------------------------------------------------------------------- regression head
def disparity_regression(x, maxdisp, keepdim=False):
    """GwcNet/submodule.py:23-27 (keepdim=False); PSMNet/submodule.py:46-54 and
    disparity_estimators/__init__.py:7-10 (keepdim=True)."""
    assert x.dim() == 4
    disp = torch.arange(maxdisp, dtype=x.dtype, device=x.device).view(1, maxdisp, 1, 1)
    return torch.sum(x * disp, 1, keepdim=keepdim)


def argmax_disparity_estimator(x, maxdisp=192):
    """disparity_estimators/__init__.py:13-15."""
    return torch.argmax(x, 1, keepdim=True)


def _mode_bounds(x):
    """Support of the mode around the arg-max of x [N,D,H,W] along D: (index, index_l, index_r), int64 [N,1,H,W].
    index_r = (first j > index with x[j] > x[j-1], the volume being extended by 1.0 at j = D) - 1;
    index_l = last j <= index with x[j] < x[j-1] (x[-1] = 1.0); D-1 / -1 when no such j exists
    (disparity_estimators/unimodal_disparity_estimator.py:6-19, dominant_modal_disparity_estimator.py:8-19)."""
    N, D, H, W = x.shape
    index = torch.argmax(x, 1, keepdim=True)
    one = x.new_ones(N, 1, H, W)
    ext = torch.cat((one, x, one), 1)
    diff = ext[:, 1:] - ext[:, :-1]                               # diff[j] = x[j] - x[j-1], j = 0..D
    j = torch.arange(D + 1, device=x.device).view(1, D + 1, 1, 1)
    rising = ((diff > 0) & (j > index)).int()
    index_r = torch.argmax(rising, 1, keepdim=True) - 1
    falling = ((diff[:, :D] < 0) & (j[:, :D] <= index)).int()
    index_l = (D - 1) - torch.argmax(torch.flip(falling, [1]), 1, keepdim=True)
    return index, index_l, index_r


def _range_mask(D, lo, hi, device):
    d = torch.arange(D, device=device).view(1, D, 1, 1)
    return (d >= lo) & (d <= hi)


def unimodal_disparity_estimator(x, maxdisp=192):
    """disparity_estimators/unimodal_disparity_estimator.py:4-25: expectation over the mode that contains the
    arg-max, re-normalised.  -> [B,1,H,W]."""
    _, lo, hi = _mode_bounds(x)
    xs = x * _range_mask(maxdisp, lo, hi, x.device)
    xs = xs / torch.sum(xs, 1, keepdim=True)
    disp = torch.arange(maxdisp, dtype=x.dtype, device=x.device).view(1, maxdisp, 1, 1)
    return torch.sum(xs * disp, 1, keepdim=True)


def _modal_mask(x):
    """dominant_modal_disparity_estimator.py:5-32: the mode's support, symmetrised around the arg-max when the
    arg-max is off-centre by 3 or more bins."""
    D = x.shape[1]
    index, lo, hi = _mode_bounds(x)
    r = torch.min(hi - index, index - lo)
    centred = torch.abs(2 * index - hi - lo) < 3
    m1 = _range_mask(D, lo, hi, x.device)
    m2 = _range_mask(D, index - r, index + r, x.device)
    return torch.where(centred, m1, m2)


def split_mode(x, maxdisp=192):
    """loss_functions/split_mode.py:9-35: (mode, mask) -- the modal mask of the RAW volume and x * mask."""
    assert x.shape[1] == maxdisp
    mask = _modal_mask(x)
    return x * mask, mask


def dominant_modal_disparity_estimator(x, maxdisp=192):
    """disparity_estimators/dominant_modal_disparity_estimator.py:35-54: 5-tap box blur along D, the blurred
    volume's main mode and its second mode (main mode removed); keep whichever holds more probability mass."""
    N, D, H, W = x.shape
    xb = x.permute(0, 2, 3, 1).reshape(N, H * W, D)
    kernel = torch.ones(H * W, 1, 5, dtype=x.dtype, device=x.device) / 5
    xb = F.conv1d(xb, kernel, padding="same", groups=H * W)
    xb = xb.permute(0, 2, 1).reshape(x.shape)
    m = _modal_mask(xb)
    y = x * m
    z = (x - y) * _modal_mask(xb * (~m))
    first = (torch.sum(y, 1) >= torch.sum(z, 1)).to(torch.float32).unsqueeze(1)
    xs = first * y + (1 - first) * z
    xs = xs / torch.sum(xs, 1, keepdim=True)
    disp = torch.arange(maxdisp, dtype=x.dtype, device=x.device).view(1, maxdisp, 1, 1)
    return torch.sum(xs * disp, 1, keepdim=True)


def regression_head(cost, maxdisp, H, W, keepdim=False, align_corners=False):
    """upsample(trilinear) -> squeeze -> softmax(dim=1) -> disparity_regression
    (GwcNet/gwcnet.py:219-224, PSMNet/stackhourglass.py:147-153, ACVNet/acv.py:247-251; with align_corners=True:
    PCWNet/pcwnet.py:446-470, CFNet/cfnet.py)."""
    c = F.interpolate(cost, [maxdisp, H, W], mode="trilinear", align_corners=align_corners)
    c = torch.squeeze(c, 1)
    p = F.softmax(c, dim=1)
    return disparity_regression(p, maxdisp, keepdim=keepdim)


# ----------------------------------------------------------------------------- conv/BN blocks
class Ctx:
    """Carries the state-dict, the train/eval flag and the BN running-stat updates."""

    def __init__(self, sd, training=False):
        self.sd = sd
        self.training = training
        self.new_stats = {}

    def bn(self, x, prefix):
        sd = self.sd
        # a layer called twice (left/right feature pass) chains its running-stat updates
        rm = self.new_stats.get(prefix + ".running_mean", sd[prefix + ".running_mean"]).clone()
        rv = self.new_stats.get(prefix + ".running_var", sd[prefix + ".running_var"]).clone()
        y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], self.training, MOMENTUM, EPS)
        if self.training:
            self.new_stats[prefix + ".running_mean"] = rm
            self.new_stats[prefix + ".running_var"] = rv
        return y


def convbn_3d(cx, x, prefix, stride=1, pad=1):
    """GwcNet/submodule.py:17-20: Conv3d(bias=False) + BatchNorm3d; `prefix`.0 conv, `prefix`.1 BN."""
    y = F.conv3d(x, cx.sd[prefix + ".0.weight"], None, stride, pad)
    return cx.bn(y, prefix + ".1")


def deconvbn_3d(cx, x, prefix):
    """ConvTranspose3d(k3, s2, p1, op1, bias=False) + BatchNorm3d (GwcNet/gwcnet.py:84-90)."""
    y = F.conv_transpose3d(x, cx.sd[prefix + ".0.weight"], None, stride=2, padding=1, output_padding=1)
    return cx.bn(y, prefix + ".1")


def _w2d(w):
    """2-D conv weight in plain NCHW memory order, whatever format the state-dict tensor is stored in (the product keeps
    its Conv2d weights channels_last; the reference -- and the fixtures generated from it -- run contiguous weights, and
    the CPU convolution picks a different kernel, with different fp32 rounding, for a channels_last weight)."""
    return w.contiguous()


def convbn_2d(cx, x, prefix, stride, pad, dilation):
    """GwcNet/submodule.py:11-14."""
    y = F.conv2d(x, _w2d(cx.sd[prefix + ".0.weight"]), None, stride, dilation if dilation > 1 else pad, dilation)
    return cx.bn(y, prefix + ".1")


def dres0(cx, x, p="dres0"):
    """GwcNet/gwcnet.py:124-127: 2 x (convbn_3d + ReLU)."""
    x = F.relu(convbn_3d(cx, x, p + ".0"))
    return F.relu(convbn_3d(cx, x, p + ".2"))


def dres1(cx, x, p="dres1"):
    """GwcNet/gwcnet.py:129-131: convbn_3d + ReLU + convbn_3d (residual added by the caller)."""
    x = F.relu(convbn_3d(cx, x, p + ".0"))
    return convbn_3d(cx, x, p + ".2")


def classif(cx, x, p):
    """GwcNet/gwcnet.py:139-153: convbn_3d + ReLU + Conv3d(32->1)."""
    x = F.relu(convbn_3d(cx, x, p + ".0"))
    return F.conv3d(x, cx.sd[p + ".2.weight"], None, 1, 1)


def hourglass_gwc(cx, x, p, attention=None):
    """GwcNet/gwcnet.py:68-105 (ACVNet/acv.py:56-93 adds `attention` after conv4)."""
    c1 = F.relu(convbn_3d(cx, x, p + ".conv1.0", stride=2))
    c2 = F.relu(convbn_3d(cx, c1, p + ".conv2.0"))
    c3 = F.relu(convbn_3d(cx, c2, p + ".conv3.0", stride=2))
    c4 = F.relu(convbn_3d(cx, c3, p + ".conv4.0"))
    if attention is not None:
        c4 = attention(cx, c4, p + ".attention_block")
    c5 = F.relu(deconvbn_3d(cx, c4, p + ".conv5") + convbn_3d(cx, c2, p + ".redir2", pad=0))
    c6 = F.relu(deconvbn_3d(cx, c5, p + ".conv6") + convbn_3d(cx, x, p + ".redir1", pad=0))
    return c6


def hourglass_psm(cx, x, presqu, postsqu, p):
    """PSMNet/stackhourglass.py:10-50."""
    out = F.relu(convbn_3d(cx, x, p + ".conv1.0", stride=2))
    pre = convbn_3d(cx, out, p + ".conv2")
    pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
    out = F.relu(convbn_3d(cx, pre, p + ".conv3.0", stride=2))
    out = F.relu(convbn_3d(cx, out, p + ".conv4.0"))
    if presqu is not None:
        post = F.relu(deconvbn_3d(cx, out, p + ".conv5") + presqu)
    else:
        post = F.relu(deconvbn_3d(cx, out, p + ".conv5") + pre)
    out = deconvbn_3d(cx, post, p + ".conv6")
    return out, pre, post


def attention_block(cx, x, p, num_heads=16, block=(4, 4, 4)):
    """ACVNet/submodule.py:383-429 (windowed 3-D self attention)."""
    sd = cx.sd
    B, C, D, H0, W0 = x.shape
    pad_r = (block[2] - W0 % block[2]) % block[2]
    pad_b = (block[1] - H0 % block[1]) % block[1]
    x = F.pad(x, (0, pad_r, 0, pad_b))
    B, C, D, H, W = x.shape
    d, h, w = D // block[0], H // block[1], W // block[2]
    x = x.view(B, C, d, block[0], h, block[1], w, block[2]).permute(0, 2, 4, 6, 3, 5, 7, 1)
    qkv = F.linear(x, sd[p + ".qkv_3d.weight"], sd[p + ".qkv_3d.bias"])
    nb = block[0] * block[1] * block[2]
    qkv = qkv.reshape(B, d * h * w, nb, 3, num_heads, C // num_heads).permute(3, 0, 1, 4, 2, 5)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * ((C // num_heads) ** -0.5)
    if pad_r > 0 or pad_b > 0:
        mask = torch.zeros((1, H, W), device=x.device)
        mask[:, -pad_b:, :].fill_(1)
        mask[:, :, -pad_r:].fill_(1)
        mask = mask.reshape(1, h, block[1], w, block[2]).transpose(2, 3).reshape(1, h * w, block[1] * block[2])
        am = mask.unsqueeze(2) - mask.unsqueeze(3)
        am = am.masked_fill(am != 0, float(-1000.0)).masked_fill(am == 0, float(0.0))
        attn = attn + am.repeat(1, d, block[0], block[0]).unsqueeze(2)
    attn = torch.softmax(attn, dim=-1)
    x = (attn @ v).view(B, d, h, w, num_heads, block[0], block[1], block[2], -1).permute(0, 4, 8, 1, 5, 2, 6, 3, 7)
    x = x.reshape(B, C, D, H, W)
    if pad_r > 0 or pad_b > 0:
        x = x[:, :, :, :H0, :W0]
    return F.conv3d(x, sd[p + ".final1x1.weight"], sd[p + ".final1x1.bias"])


# ----------------------------------------------------------------------------- 2-D feature CNNs
def _basic_block(cx, x, p, stride, pad, dilation, has_down):
    out = F.relu(convbn_2d(cx, x, p + ".conv1.0", stride, pad, dilation))
    out = convbn_2d(cx, out, p + ".conv2", 1, pad, dilation)
    if has_down:
        x = convbn_2d(cx, x, p + ".downsample", stride, 0, 1)
    return out + x


def _trunk(cx, x, p):
    """firstconv + layer1..4 shared by the three models (GwcNet/gwcnet.py:17-58)."""
    x = F.relu(convbn_2d(cx, x, p + ".firstconv.0", 2, 1, 1))
    x = F.relu(convbn_2d(cx, x, p + ".firstconv.2", 1, 1, 1))
    x = F.relu(convbn_2d(cx, x, p + ".firstconv.4", 1, 1, 1))
    for i in range(3):
        x = _basic_block(cx, x, f"{p}.layer1.{i}", 1, 1, 1, False)
    l2 = x
    for i in range(16):
        l2 = _basic_block(cx, l2, f"{p}.layer2.{i}", 2 if i == 0 else 1, 1, 1, i == 0)
    l3 = l2
    for i in range(3):
        l3 = _basic_block(cx, l3, f"{p}.layer3.{i}", 1, 1, 1, i == 0)
    l4 = l3
    for i in range(3):
        l4 = _basic_block(cx, l4, f"{p}.layer4.{i}", 1, 1, 2, False)
    return l2, l3, l4


def features_gwc(cx, x, concat, p="feature_extraction"):
    """GwcNet/gwcnet.py:12-65."""
    l2, l3, l4 = _trunk(cx, x, p)
    gwc = torch.cat((l2, l3, l4), dim=1)
    if not concat:
        return gwc, None
    c = F.relu(convbn_2d(cx, gwc, p + ".lastconv.0", 1, 1, 1))
    c = F.conv2d(c, _w2d(cx.sd[p + ".lastconv.2.weight"]))
    return gwc, c


def features_psm(cx, x, p="feature_extraction"):
    """PSMNet/submodule.py:57-132 (SPP)."""
    l2, l3, l4 = _trunk(cx, x, p)
    size = (l4.shape[2], l4.shape[3])
    branches = []
    for name, k in (("branch1", 64), ("branch2", 32), ("branch3", 16), ("branch4", 8)):
        b = F.avg_pool2d(l4, (k, k), stride=(k, k))
        b = F.relu(convbn_2d(cx, b, f"{p}.{name}.1", 1, 0, 1))
        branches.append(F.interpolate(b, size, mode="bilinear", align_corners=False))
    feat = torch.cat((l2, l4, branches[3], branches[2], branches[1], branches[0]), 1)
    f = F.relu(convbn_2d(cx, feat, p + ".lastconv.0", 1, 1, 1))
    return F.conv2d(f, _w2d(cx.sd[p + ".lastconv.2.weight"]))


# ----------------------------------------------------------------------------- whole models
def feature_noise(eps, seed):
    """Test helper: a `feature_hook` that multiplies the k-th feature map it sees by (1 + eps * N(0,1)) -- a relative
    perturbation of the size of fp32 rounding / of the measured distance between two fp32 evaluations of the 2-D CNN.
    Differentiable, so the 2-D CNN's parameter gradients respond too.  Used by the train-parity tests to measure how far
    the EXACT (fp64) gradients of a test configuration move under rounding-sized changes (tests/test_models.py)."""
    state = {"k": 0}

    def hook(t):
        g = torch.Generator().manual_seed(1000 * seed + state["k"])
        state["k"] += 1
        return t * (1 + eps * torch.randn(t.shape, generator=g, dtype=torch.float64).to(t.dtype))
    return hook


def gwcnet_forward(sd, left, right, maxdisp, use_concat_volume, training=False, return_ctx=False, feature_hook=None):
    """GwcNet/gwcnet.py:171-224.  `feature_hook` (tests): applied to every 1/4-resolution feature map (see feature_noise)."""
    cx = Ctx(sd, training)
    gl, cl = features_gwc(cx, left, use_concat_volume)
    gr, cr = features_gwc(cx, right, use_concat_volume)
    if feature_hook is not None:
        gl, gr = feature_hook(gl), feature_hook(gr)
        if cl is not None:
            cl, cr = feature_hook(cl), feature_hook(cr)
    out = gwcnet_aggregate(cx, gl, gr, cl, cr, maxdisp, left.shape[2], left.shape[3])
    return (out, cx) if return_ctx else out


def gwcnet_aggregate(cx, gl, gr, cl, cr, maxdisp, H, W):
    """GwcNet/gwcnet.py:175-224 from the 1/4-resolution features onwards (gl / gr: 320-channel gwc features, cl / cr: the
    12-channel concat features or None) -- the hand-written part of the product, separable for error attribution."""
    vol = build_gwc_volume(gl, gr, maxdisp // 4, 40)
    if cl is not None:
        vol = torch.cat((vol, build_concat_volume(cl, cr, maxdisp // 4)), 1)
    cost0 = dres0(cx, vol)
    cost0 = dres1(cx, cost0) + cost0
    out1 = hourglass_gwc(cx, cost0, "dres2")
    out2 = hourglass_gwc(cx, out1, "dres3")
    out3 = hourglass_gwc(cx, out2, "dres4")
    if cx.training:
        return [regression_head(classif(cx, o, f"classif{i}"), maxdisp, H, W)
                for i, o in enumerate((cost0, out1, out2, out3))]
    return regression_head(classif(cx, out3, "classif3"), maxdisp, H, W)


def psmnet_forward(sd, left, right, maxdisp, training=False, return_ctx=False):
    """PSMNet/stackhourglass.py:103-161."""
    cx = Ctx(sd, training)
    fl = features_psm(cx, left)
    fr = features_psm(cx, right)
    out = psmnet_aggregate(cx, fl, fr, maxdisp, left.shape[2], left.shape[3])
    return (out, cx) if return_ctx else out


def psmnet_aggregate(cx, fl, fr, maxdisp, H, W):
    """PSMNet/stackhourglass.py:111-161 from the 32-channel features onwards."""
    training = cx.training
    cost = build_concat_volume(fl, fr, maxdisp // 4)
    cost0 = dres0(cx, cost)
    cost0 = dres1(cx, cost0) + cost0
    out1, pre1, post1 = hourglass_psm(cx, cost0, None, None, "dres2")
    out1 = out1 + cost0
    out2, pre2, post2 = hourglass_psm(cx, out1, pre1, post1, "dres3")
    out2 = out2 + cost0
    out3, pre3, post3 = hourglass_psm(cx, out2, pre1, post2, "dres4")
    out3 = out3 + cost0
    cost1 = classif(cx, out1, "classif1")
    cost2 = classif(cx, out2, "classif2") + cost1
    cost3 = classif(cx, out3, "classif3") + cost2
    pred3 = regression_head(cost3, maxdisp, H, W, keepdim=True)
    if training:
        pred1 = regression_head(cost1, maxdisp, H, W, keepdim=True)
        pred2 = regression_head(cost2, maxdisp, H, W, keepdim=True)
        return [pred1, pred2, pred3]
    return pred3


def acv_patch_volume(sd, gwc_volume):
    """ACVNet/acv.py:109-112,183-187: depth-wise (1,3,3) convs, dilations 1 / 1,2,3."""
    v = F.conv3d(gwc_volume, sd["patch.weight"], None, 1, (0, 1, 1), 1, 40)
    p1 = F.conv3d(v[:, :8], sd["patch_l1.weight"], None, 1, (0, 1, 1), 1, 8)
    p2 = F.conv3d(v[:, 8:24], sd["patch_l2.weight"], None, 1, (0, 2, 2), 2, 16)
    p3 = F.conv3d(v[:, 24:40], sd["patch_l3.weight"], None, 1, (0, 3, 3), 3, 16)
    return torch.cat((p1, p2, p3), dim=1)


def acvnet_forward(sd, left, right, maxdisp, attn_weights_only=False, freeze_attn_weights=False,
                   training=False, return_ctx=False, feature_hook=None):
    """ACVNet/acv.py:162-253.  `feature_hook` (tests): applied to both gwc feature maps (see feature_noise)."""
    cx = Ctx(sd, training)
    H, W = left.shape[2], left.shape[3]
    if freeze_attn_weights:       # acv.py:164-176: the feature CNN belongs to the frozen attention branch
        with torch.no_grad():
            gl, _ = features_gwc(cx, left, False)
            gr, _ = features_gwc(cx, right, False)
    else:
        gl, _ = features_gwc(cx, left, False)
        gr, _ = features_gwc(cx, right, False)
    if feature_hook is not None:
        gl, gr = feature_hook(gl), feature_hook(gr)
    out = acvnet_aggregate(cx, gl, gr, maxdisp, H, W, attn_weights_only, freeze_attn_weights)
    return (out, cx) if return_ctx else out


def acv_concat_features(cx, g):
    """ACVNet/acv.py:104-107,192-194: `concatconv` on the 320-channel gwc feature."""
    c = F.relu(convbn_2d(cx, g, "concatconv.0", 1, 1, 1))
    return F.conv2d(c, _w2d(cx.sd["concatconv.2.weight"]))


def acvnet_aggregate(cx, gl, gr, maxdisp, H, W, attn_weights_only=False, freeze_attn_weights=False, cl=None, cr=None):
    """ACVNet/acv.py:166-253 behind the feature extractor: everything from the 1/4-resolution gwc features on.  `cl` / `cr`
    (optional) replace the `concatconv` outputs -- the cut used by the isolation tests, which hand BOTH implementations the
    same 2-D features."""
    sd, training = cx.sd, cx.training

    def att_branch():
        gwc = build_gwc_volume(gl, gr, maxdisp // 4, 40)
        pv = acv_patch_volume(sd, gwc)
        ca = dres1(cx, pv, "dres1_att_")
        ca = hourglass_gwc(cx, ca, "dres2_att_", attention=attention_block)
        return classif(cx, ca, "classif_att_")

    if freeze_attn_weights:
        with torch.no_grad():
            att = att_branch()
    else:
        att = att_branch()

    if not attn_weights_only:
        if cl is None:
            cl, cr = acv_concat_features(cx, gl), acv_concat_features(cx, gr)
        cvol = build_concat_volume(cl, cr, maxdisp // 4, mask_left=False)
        ac = F.softmax(att, dim=2) * cvol
        cost0 = dres0(cx, ac)
        cost0 = dres1(cx, cost0) + cost0
        out1 = hourglass_gwc(cx, cost0, "dres2", attention=attention_block)
        out2 = hourglass_gwc(cx, out1, "dres3", attention=attention_block)

    if training:
        preds = []
        if not freeze_attn_weights:
            preds.append(regression_head(att, maxdisp, H, W))
        if not attn_weights_only:
            preds += [regression_head(classif(cx, o, f"classif{i}"), maxdisp, H, W)
                      for i, o in enumerate((cost0, out1, out2))]
        return preds
    if attn_weights_only:
        return regression_head(att, maxdisp, H, W)
    return regression_head(classif(cx, out2, "classif2"), maxdisp, H, W)


# ----------------------------------------------------------------------------- loss used by bench/tests
# ----------------------------------------------------------------------------- PCWNet
def mish(x):
    """PCWNet/submodule.py:11-18,178-190: x * tanh(softplus(x))."""
    return x * torch.tanh(F.softplus(x))


def _pcw_block(cx, x, p, stride, pad, dilation, has_down):
    """PCWNet/submodule.py:192-215: BasicBlock with Mish after the first conv only."""
    out = mish(convbn_2d(cx, x, p + ".conv1.0", stride, pad, dilation))
    out = convbn_2d(cx, out, p + ".conv2", 1, pad, dilation)
    if has_down:
        x = convbn_2d(cx, x, p + ".downsample", stride, 0, 1)
    return out + x


def _pcw_layer(cx, x, p, blocks, stride, pad, dilation, has_down):
    for i in range(blocks):
        x = _pcw_block(cx, x, f"{p}.{i}", stride if i == 0 else 1, pad, dilation, has_down and i == 0)
    return x


def _pcw_head2d(cx, x, p):
    """convbn(3x3) + Mish + 1x1 Conv2d (pcwnet.py:36-74)."""
    return F.conv2d(mish(convbn_2d(cx, x, p + ".0", 1, 1, 1)), _w2d(cx.sd[p + ".2.weight"]))


def features_pcw(cx, x, p="feature_extraction"):
    """PCWNet/pcwnet.py:12-131 with concat_feature=True."""
    x = mish(convbn_2d(cx, x, p + ".firstconv.0", 2, 1, 1))
    x = mish(convbn_2d(cx, x, p + ".firstconv.2", 1, 1, 1))
    x = mish(convbn_2d(cx, x, p + ".firstconv.4", 1, 1, 1))
    x = _pcw_layer(cx, x, p + ".layer1", 3, 1, 1, 1, False)
    l2 = _pcw_layer(cx, x, p + ".layer2", 16, 2, 1, 1, True)
    l3 = _pcw_layer(cx, l2, p + ".layer3", 3, 1, 1, 1, True)
    l4 = _pcw_layer(cx, l3, p + ".layer4", 3, 1, 1, 2, False)
    l5 = _pcw_layer(cx, l4, p + ".layer5", 3, 2, 1, 1, True)
    l6 = _pcw_layer(cx, l5, p + ".layer7", 3, 2, 1, 1, True)
    l7 = _pcw_layer(cx, l6, p + ".layer9", 3, 2, 1, 1, True)
    comb = torch.cat((l2, l3, l4), 1)
    refine = mish(convbn_2d(cx, comb, p + ".layer_refine.0", 1, 1, 1))
    refine = mish(convbn_2d(cx, refine, p + ".layer_refine.2", 1, 0, 1))
    return {"gw1": _pcw_head2d(cx, comb, p + ".layer11"), "gw2": _pcw_head2d(cx, l5, p + ".gw2"),
            "gw3": _pcw_head2d(cx, l6, p + ".gw3"), "gw4": _pcw_head2d(cx, l7, p + ".gw4"),
            "concat_feature1": _pcw_head2d(cx, comb, p + ".lastconv"), "finetune_feature": refine,
            "concat_feature2": _pcw_head2d(cx, l5, p + ".concat2"), "concat_feature3": _pcw_head2d(cx, l6, p + ".concat3"),
            "concat_feature4": _pcw_head2d(cx, l7, p + ".concat4")}


def hourglass_pcw(cx, x, p):
    """PCWNet/pcwnet.py:211-252."""
    c1 = mish(convbn_3d(cx, x, p + ".conv1.0", 2, 1))
    c2 = mish(convbn_3d(cx, c1, p + ".conv2.0", 1, 1))
    c3 = mish(convbn_3d(cx, c2, p + ".conv3.0", 2, 1))
    c4 = mish(convbn_3d(cx, c3, p + ".conv4.0", 1, 1))
    c5 = mish(deconvbn_3d(cx, c4, p + ".conv5") + convbn_3d(cx, c2, p + ".redir2", 1, 0))
    return mish(deconvbn_3d(cx, c5, p + ".conv6") + convbn_3d(cx, x, p + ".redir1", 1, 0))


def hourglassup_pcw(cx, x, f4, f5, f6, p):
    """PCWNet/pcwnet.py:133-208."""
    sd = cx.sd
    c1 = F.conv3d(x, sd[p + ".conv1.weight"], None, 2, 1)
    c1 = mish(convbn_3d(cx, torch.cat((c1, f4), 1), p + ".combine1.0", 1, 1))
    c2 = mish(convbn_3d(cx, c1, p + ".conv2.0", 1, 1))
    c3 = F.conv3d(c2, sd[p + ".conv3.weight"], None, 2, 1)
    c3 = mish(convbn_3d(cx, torch.cat((c3, f5), 1), p + ".combine2.0", 1, 1))
    c4 = mish(convbn_3d(cx, c3, p + ".conv4.0", 1, 1))
    c5 = F.conv3d(c4, sd[p + ".conv5.weight"], None, 2, 1)
    c5 = mish(convbn_3d(cx, torch.cat((c5, f6), 1), p + ".combine3.0", 1, 1))
    c6 = mish(convbn_3d(cx, c5, p + ".conv6.0", 1, 1))
    c7 = mish(deconvbn_3d(cx, c6, p + ".conv7") + convbn_3d(cx, c4, p + ".redir3", 1, 0))
    c8 = mish(deconvbn_3d(cx, c7, p + ".conv8") + convbn_3d(cx, c2, p + ".redir2", 1, 0))
    return mish(deconvbn_3d(cx, c8, p + ".conv9") + convbn_3d(cx, x, p + ".redir1", 1, 0))


def pcw_correlation_volume(ref, tgt, maxdisp, num_groups):
    """PCWNet/submodule.py:121-135 (`build_corrleation_volume`), literal semantics: slice i+maxdisp holds, for i >= 0,
    the correlation of ref[w] with tgt[w-i] at w >= i; for i < 0 the FIRST |i| reference columns against the LAST |i|
    target columns (the reference's `[..., :-i]` with negative i)."""
    B, C, H, W = ref.shape
    cpg = C // num_groups
    vol = ref.new_zeros(B, num_groups, 2 * maxdisp + 1, H, W)
    for i in range(-maxdisp, maxdisp + 1):
        if i >= 0:
            prod = ref[..., i:] * tgt[..., :W - i]
            vol[:, :, i + maxdisp, :, i:] = prod.view(B, num_groups, cpg, H, W - i).mean(2)
        else:
            prod = ref[..., :-i] * tgt[..., W + i:]
            vol[:, :, i + maxdisp, :, :-i] = prod.view(B, num_groups, cpg, H, -i).mean(2)
    return vol


def pcw_warp(x, disp):
    """PCWNet/submodule.py:137-176: bilinear sampling of x at column w - disp (grid normalised with W-1/H-1, then
    grid_sample's align_corners=False), zeroed where the sampling footprint left the image."""
    B, C, H, W = x.shape
    xx = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
    yy = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1).expand(B, 1, H, W)
    grid = torch.cat((2.0 * (xx - disp) / max(W - 1, 1) - 1.0, 2.0 * yy / max(H - 1, 1) - 1.0), 1).permute(0, 2, 3, 1)
    out = F.grid_sample(x, grid, align_corners=False)
    mask = F.grid_sample(torch.ones_like(x), grid, align_corners=False)
    mask = torch.where(mask < 0.999, torch.zeros_like(mask), torch.ones_like(mask))
    return out * mask


def _pcw_refine(cx, x, disp, p="refinenet3"):
    """PCWNet/pcwnet.py:254-308."""
    x = mish(convbn_2d(cx, x, p + ".conv1.0", 1, 1, 1))
    x = mish(convbn_2d(cx, x, p + ".conv2.0", 1, 1, 1))
    x = mish(convbn_2d(cx, x, p + ".conv3.0", 1, 2, 2))
    x = mish(convbn_2d(cx, x, p + ".conv4.0", 1, 4, 4))
    x = _pcw_block(cx, x, p + ".conv5.0", 1, 1, 8, True)
    x = _pcw_block(cx, x, p + ".conv6.0", 1, 1, 16, True)
    x = _pcw_block(cx, x, p + ".conv7.0", 1, 1, 1, True)
    return disp + F.conv2d(x, _w2d(cx.sd[p + ".conv8.weight"]), None, 1, 1)


def pcwnet_forward(sd, left, right, maxdisp, training=False, return_ctx=False):
    """PCWNet/pcwnet.py:384-511 (PCWNet_GC).  train: [pred0, combine, pred1, pred2, pred3, disp_finetune]."""
    cx = Ctx(sd, training)
    fl, fr = features_pcw(cx, left), features_pcw(cx, right)
    vols = []
    for k, div in ((1, 4), (2, 8), (3, 16), (4, 32)):
        g = build_gwc_volume(fl[f"gw{k}"], fr[f"gw{k}"], maxdisp // div, 40)
        c = build_concat_volume(fl[f"concat_feature{k}"], fr[f"concat_feature{k}"], maxdisp // div)
        vols.append(torch.cat((g, c), 1))
    x = mish(convbn_3d(cx, vols[0], "dres0.0"))
    cost0 = mish(convbn_3d(cx, x, "dres0.2"))
    cost0 = convbn_3d(cx, mish(convbn_3d(cx, cost0, "dres1.0")), "dres1.2") + cost0
    combine = hourglassup_pcw(cx, cost0, vols[1], vols[2], vols[3], "combine1")
    out1 = hourglass_pcw(cx, combine, "dres2")
    out2 = hourglass_pcw(cx, out1, "dres3")
    out3 = hourglass_pcw(cx, out2, "dres4")
    H, W = left.shape[2], left.shape[3]

    def head(xx, p):
        c = F.conv3d(mish(convbn_3d(cx, xx, p + ".0")), sd[p + ".2.weight"], None, 1, 1)
        return regression_head(c, maxdisp, H, W, align_corners=True)

    preds = None
    if training:        # the reference evaluates classif0..3 then classif4 (BN running-stat order is irrelevant: distinct layers)
        preds = [head(cost0, "classif0"), head(out1, "classif1"), head(out2, "classif2")]
    pred3 = head(out3, "classif3")
    comb_pred = head(combine, "classif4") if training else None
    p3 = pred3.unsqueeze(1)
    rl = F.interpolate(fl["finetune_feature"], [H, W], mode="bilinear", align_corners=True)
    rr = F.interpolate(fr["finetune_feature"], [H, W], mode="bilinear", align_corners=True)
    rw = pcw_warp(rr, p3)
    corr = pcw_correlation_volume(rl, rw, 24, 1).squeeze(1)
    up = mish(convbn_2d(cx, p3, "dispupsample.0", 1, 0, 1))
    fine = _pcw_refine(cx, torch.cat((rl - rw, rl, up, p3, corr), 1), p3).squeeze(1)
    if training:
        out = [preds[0], comb_pred, preds[1], preds[2], pred3, fine]
        return (out, cx) if return_ctx else out
    return (fine, cx) if return_ctx else fine


def smooth_l1_multi(preds, gt, maxdisp, weights):
    """Supervised loss for the train step (the reference ships none for these models; weights
    follow the GwcNet paper).  Valid mask: trainer/trainer_torchrun.py:272."""
    mask = (gt > 0) & (gt < maxdisp - 1)
    loss = 0.0
    for p, w in zip(preds, weights):
        p = p.squeeze(1) if p.dim() == 4 else p
        loss = loss + w * F.smooth_l1_loss(p[mask], gt[mask], reduction="mean")
    return loss


# ----------------------------------------------------------------------------- evaluation-path input step
def pad_to_2x(left, right, disp=None, mask=None, scale=96):
    """datasets/data_augmentation/__init__.py:57-80 on torch tensors: left/right [H,W,C] zero-padded on TOP and to the
    RIGHT up to multiples of 96; disp [H,W] (disparity) or [D,H,W] (distribution) and mask [H,W] likewise."""
    H, W, _ = left.shape
    top = int(math.ceil(H / scale) * scale - H)
    rp = int(math.ceil(W / scale) * scale - W)
    left = F.pad(left, (0, 0, 0, rp, top, 0))
    right = F.pad(right, (0, 0, 0, rp, top, 0))
    if disp is not None:
        disp = F.pad(disp, (0, rp, top, 0))
    if mask is not None:
        assert disp.dim() == 2
        mask = F.pad(mask, (0, rp, top, 0))
    return left, right, disp, mask


def to_tensor_normalize(img_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """datasets/utils.py:62-69 `get_transform()` = torchvision.transforms.ToTensor + Normalize.  torchvision is a
    third-party dependency of the reference (setup.py:24, no version pinned) that is absent from this image; its
    published algorithm is restated: ToTensor = HWC uint8 -> CHW float32 `.div(255)`; Normalize =
    `tensor.sub_(mean[:,None,None]).div_(std[:,None,None])` with mean / std as float32 tensors.  [H,W,3] -> [3,H,W]."""
    t = img_u8.permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return t.sub_(m).div_(s)


def prepare_view(img_u8):
    """pad_to_2x followed by get_transform on one uint8 [H,W,3] view (what the datasets' test branch does)."""
    padded, _, _, _ = pad_to_2x(img_u8, img_u8)
    return to_tensor_normalize(padded)


# ----------------------------------------------------------------------------- IGEV-family initial volume
def igev_init_volume(match_left, match_right, maxdisp):
    """IGEVStereo/igev_stereo.py:206: `build_gwc_volume(match_left, match_right, max_disp // 4, 8)` on the 96-channel
    matching features (12 channels per group); IGEVStereo/submodule.py:153-171 is the same arithmetic as GwcNet's."""
    return build_gwc_volume(match_left, match_right, maxdisp // 4, 8)


def igev_init_disparity(cost, maxdisp):
    """igev_stereo.py:211-212: prob = softmax(classifier(volume).squeeze(1), dim=1); init_disp =
    disparity_regression(prob, max_disp // 4) at 1/4 resolution, keepdim (IGEVStereo/submodule.py:221-225)."""
    prob = F.softmax(cost.squeeze(1), dim=1)
    return disparity_regression(prob, maxdisp // 4, keepdim=True)


def igev_basic_conv(cx, x, p, is_3d=True, deconv=False, bn=True, relu=True, stride=1, pad=1):
    """IGEVStereo/submodule.py:9-38 `BasicConv`: conv / transposed conv (bias=False) [+ BatchNorm] [+ LeakyReLU()].
    Keys p.conv.weight, p.bn.*; deconv = ConvTranspose3d(k4, s2, p1) (igev_stereo.py:44-51)."""
    w = cx.sd[p + ".conv.weight"]
    if is_3d:
        x = F.conv_transpose3d(x, w, None, stride=2, padding=1) if deconv else F.conv3d(x, w, None, stride, pad)
    else:
        x = F.conv2d(x, _w2d(w), None, stride, pad)
    if bn:
        x = cx.bn(x, p + ".bn")
    return F.leaky_relu(x, 0.01) if relu else x


def igev_feature_att(cx, cv, feat, p):
    """IGEVStereo/submodule.py:228-241 `FeatureAtt`: sigmoid(Conv2d(1x1, bias)(BasicConv2d(1x1)(feat))).unsqueeze(2) * cv."""
    a = igev_basic_conv(cx, feat, p + ".feat_att.0", is_3d=False, pad=0)
    a = F.conv2d(a, _w2d(cx.sd[p + ".feat_att.1.weight"]), cx.sd[p + ".feat_att.1.bias"])
    return torch.sigmoid(a.unsqueeze(2)) * cv


def igev_hourglass(cx, x, features, p):
    """IGEVStereo/igev_stereo.py:23-100 `hourglass(in_channels)` forward (:76-100)."""
    def seq2(t, q, s):
        t = igev_basic_conv(cx, t, q + ".0", stride=s)
        return igev_basic_conv(cx, t, q + ".1")

    def agg(t, q):
        t = igev_basic_conv(cx, t, q + ".0", pad=0)
        t = igev_basic_conv(cx, t, q + ".1")
        return igev_basic_conv(cx, t, q + ".2")
    conv1 = igev_feature_att(cx, seq2(x, p + ".conv1", 2), features[1], p + ".feature_att_8")
    conv2 = igev_feature_att(cx, seq2(conv1, p + ".conv2", 2), features[2], p + ".feature_att_16")
    conv3 = igev_feature_att(cx, seq2(conv2, p + ".conv3", 2), features[3], p + ".feature_att_32")
    conv3_up = igev_basic_conv(cx, conv3, p + ".conv3_up", deconv=True)
    conv2 = agg(torch.cat((conv3_up, conv2), dim=1), p + ".agg_0")
    conv2 = igev_feature_att(cx, conv2, features[2], p + ".feature_att_up_16")
    conv2_up = igev_basic_conv(cx, conv2, p + ".conv2_up", deconv=True)
    conv1 = agg(torch.cat((conv2_up, conv1), dim=1), p + ".agg_1")
    conv1 = igev_feature_att(cx, conv1, features[1], p + ".feature_att_up_8")
    return igev_basic_conv(cx, conv1, p + ".conv1_up", deconv=True, bn=False, relu=False)


def igev_cost_aggregation(sd, match_left, match_right, features_left, maxdisp, training=False, return_ctx=False):
    """IGEVStereo/igev_stereo.py:206-213 with the modules of :148-151 (`corr_stem`, `corr_feature_att`, `cost_agg`,
    `classifier`): -> (geo_encoding_volume [B,8,D/4,H/4,W/4], init_disp [B,1,H/4,W/4])."""
    cx = Ctx(sd, training)
    vol = igev_init_volume(match_left, match_right, maxdisp)
    vol = igev_basic_conv(cx, vol, "corr_stem")
    vol = igev_feature_att(cx, vol, features_left[0], "corr_feature_att")
    geo = igev_hourglass(cx, vol, features_left, "cost_agg")
    cost = F.conv3d(geo, sd["classifier.weight"], None, 1, 1)
    out = (geo, igev_init_disparity(cost, maxdisp))
    return (out, cx) if return_ctx else out


# ----------------------------------------------------------------------------- CFNet
def _cf_conv_bn_mish(cx, x, p):
    """CFNet/submodule.py:70-93 `conv2DBatchNormRelu` (1x1 conv, BN, Mish): keys p.cbr_unit.{0,1}."""
    return mish(cx.bn(F.conv2d(x, _w2d(cx.sd[p + ".cbr_unit.0.weight"])), p + ".cbr_unit.1"))


def _cf_pyramid_pooling(cx, x, p):
    """CFNet/submodule.py:11-68 in the configuration cfnet.py:31 uses (pool_sizes=None, fusion 'sum', 'icnet')."""
    import numpy as np
    h, w = x.shape[2:]
    sizes = [(int(h / s), int(w / s)) for s in np.linspace(2, min(h, w), 4, dtype=int)][::-1]
    acc = x
    for i, k in enumerate(sizes):
        out = _cf_conv_bn_mish(cx, F.avg_pool2d(x, k, stride=k, padding=0), f"{p}.path_module_list.{i}")
        acc = acc + 0.25 * F.interpolate(out, size=(h, w), mode="bilinear", align_corners=False)
    return mish(acc / 2.0)


def features_cf(cx, x, p="feature_extraction"):
    """CFNet/cfnet.py:12-176 with concat_feature=True."""
    def up(t, q):            # nn.Upsample(scale_factor=2) (nearest) + convbn + Mish
        return mish(convbn_2d(cx, F.interpolate(t, scale_factor=2), q + ".1", 1, 1, 1))
    x = mish(convbn_2d(cx, x, p + ".firstconv.0", 2, 1, 1))
    x = mish(convbn_2d(cx, x, p + ".firstconv.2", 1, 1, 1))
    x = mish(convbn_2d(cx, x, p + ".firstconv.4", 1, 1, 1))
    l2 = _pcw_layer(cx, x, p + ".layer2", 1, 1, 1, 1, True)
    l3 = _pcw_layer(cx, l2, p + ".layer3", 1, 2, 1, 1, True)
    l4 = _pcw_layer(cx, l3, p + ".layer4", 1, 2, 1, 1, True)
    l5 = _pcw_layer(cx, l4, p + ".layer5", 1, 2, 1, 1, True)
    l6 = _cf_pyramid_pooling(cx, _pcw_layer(cx, l5, p + ".layer6", 1, 2, 1, 1, True), p + ".pyramid_pooling")
    d5 = mish(convbn_2d(cx, torch.cat((l5, up(l6, p + ".upconv6")), 1), p + ".iconv5.0", 1, 1, 1))
    d4 = mish(convbn_2d(cx, torch.cat((l4, up(d5, p + ".upconv5")), 1), p + ".iconv4.0", 1, 1, 1))
    d3 = mish(convbn_2d(cx, torch.cat((l3, up(d4, p + ".upconv4")), 1), p + ".iconv3.0", 1, 1, 1))
    d2 = mish(convbn_2d(cx, torch.cat((l2, up(d3, p + ".upconv3")), 1), p + ".iconv2.0", 1, 1, 1))
    feats = {"2": d2, "3": d3, "4": d4, "5": d5, "6": l6}
    out = {}
    for k, t in feats.items():
        out["gw" + k] = _pcw_head2d(cx, t, p + ".gw" + k)
        out["concat_feature" + k] = _pcw_head2d(cx, t, p + ".concat" + k)
    return out


def hourglassup_cf(cx, x, f4, f5, p):
    """CFNet/cfnet.py:178-228."""
    sd = cx.sd
    c1 = F.conv3d(x, sd[p + ".conv1.weight"], None, 2, 1)
    c1 = mish(convbn_3d(cx, torch.cat((c1, f4), 1), p + ".combine1.0", 1, 1))
    c2 = mish(convbn_3d(cx, c1, p + ".conv2.0", 1, 1))
    c3 = F.conv3d(c2, sd[p + ".conv3.weight"], None, 2, 1)
    c3 = mish(convbn_3d(cx, torch.cat((c3, f5), 1), p + ".combine2.0", 1, 1))
    c4 = mish(convbn_3d(cx, c3, p + ".conv4.0", 1, 1))
    c8 = mish(deconvbn_3d(cx, c4, p + ".conv8") + convbn_3d(cx, c2, p + ".redir2", 1, 0))
    return mish(deconvbn_3d(cx, c8, p + ".conv9") + convbn_3d(cx, x, p + ".redir1", 1, 0))


def _cf_dres(cx, x, p0, p1):
    c = mish(convbn_3d(cx, mish(convbn_3d(cx, x, p0 + ".0")), p0 + ".2"))
    return convbn_3d(cx, mish(convbn_3d(cx, c, p1 + ".0")), p1 + ".2") + c


def _cf_classif(cx, x, p):
    return F.conv3d(mish(convbn_3d(cx, x, p + ".0")), cx.sd[p + ".2.weight"], None, 1, 1).squeeze(1)


def cf_disparity_variance(x, maxdisp, disparity):
    """CFNet/submodule.py:128-134."""
    d = torch.arange(0, maxdisp, dtype=x.dtype, device=x.device).view(1, maxdisp, 1, 1)
    return torch.sum(x * (d - disparity) ** 2, 1, keepdim=True)


def cf_disparity_variance_confidence(x, disparity_samples, disparity):
    """CFNet/submodule.py:136-140."""
    return torch.sum(x * (disparity - disparity_samples) ** 2, 1, keepdim=True)


def cf_sampled_volume(left, right, samples, num_groups):
    """CFNet/submodule.py:306-350 (`SpatialTransformer`) + cfnet.py:479-497 (`cost_volume_generator`): right features
    gathered at column w - sample (clamped index; zero where the un-clamped column leaves the image), then either
    the concat of (left, warped right) (num_groups=None) or their group-wise correlation.  samples: float [B,S,H,W]."""
    B, C, H, W = left.shape
    S = samples.shape[1]
    pos = torch.arange(0.0, W).view(1, 1, 1, W) - samples
    idx = pos.clamp(0, W - 1).long().unsqueeze(1).expand(B, C, S, H, W)
    warped = torch.gather(right.unsqueeze(2).expand(B, C, S, H, W), 4, idx)
    warped = (1 - ((pos < 0) | (pos > W - 1)).float().unsqueeze(1)) * warped
    lmap = left.unsqueeze(2).expand(B, C, S, H, W)
    if num_groups is None:
        return torch.cat((lmap, warped), 1)
    return (lmap * warped).view(B, num_groups, C // num_groups, S, H, W).mean(2)


def cfnet_forward(sd, left, right, maxdisp, training=False, return_ctx=False, forced_samples=None):
    """CFNet/cfnet.py:499-666 (CFNet = cfnet(use_concat_volume=True)).  train: the 9 predictions of cfnet.py:653.
    forced_samples = (samples_s3, samples_s2) [B,S,H,W] replaces the integer disparity samples of the two cascade
    stages (tests: the samples the reference drew; see tests/golden/make_golden_cfnet.py)."""
    cx = Ctx(sd, training)
    fl, fr = features_cf(cx, left), features_cf(cx, right)
    H, W = left.shape[2], left.shape[3]
    vols = {}
    for k, div in (("4", 8), ("5", 16), ("6", 32)):
        g = build_gwc_volume(fl["gw" + k], fr["gw" + k], maxdisp // div, 40)
        c = build_concat_volume(fl["concat_feature" + k], fr["concat_feature" + k], maxdisp // div)
        vols[k] = torch.cat((g, c), 1)
    cost0_4 = _cf_dres(cx, vols["4"], "dres0", "dres1")
    cost0_5 = _cf_dres(cx, vols["5"], "dres0_5", "dres1_5")
    cost0_6 = _cf_dres(cx, vols["6"], "dres0_6", "dres1_6")
    out1_4 = hourglassup_cf(cx, cost0_4, cost0_5, cost0_6, "combine1")
    out2_4 = hourglass_pcw(cx, out1_4, "dres3")
    poss4 = F.softmax(_cf_classif(cx, out2_4, "classif2"), dim=1)
    pred2_s4 = disparity_regression(poss4, maxdisp // 8).unsqueeze(1)
    cur = pred2_s4.detach()
    dv = torch.arange(0, maxdisp // 8, dtype=left.dtype).view(1, -1, 1, 1)
    var = torch.sum(poss4 * (dv - cur) ** 2, 1, keepdim=True).sqrt()

    def up(x, scale, h, w):
        return F.interpolate(x * scale, [h, w], mode="bilinear", align_corners=True)

    def search_range(count, lo, hi, scale):
        top = maxdisp // (2 ** scale) - 1
        slack = torch.clamp(count - hi + lo, min=0) / 2.0
        return torch.clamp(lo - slack, min=0, max=top), torch.clamp(hi + slack, min=0, max=top)

    def samples_of(lo, hi, count):
        k = torch.arange(1.0, count + 1, 1).view(count, 1, 1)
        mid = lo + (hi - lo) / (count + 1) * k
        return torch.cat((torch.floor(lo), mid, torch.ceil(hi)), 1).long().float()

    def stage(k, lo, hi, count, scale, groups, tag):
        lo1, hi1 = search_range(count + 1, lo, hi, scale)
        smp = samples_of(lo1, hi1, count)
        if forced_samples is not None:
            smp = forced_samples[0 if tag == "s3" else 1].to(smp.dtype)
        v_cat = cf_sampled_volume(fl["concat_feature" + k], fr["concat_feature" + k], smp, None)
        v_gwc = cf_sampled_volume(fl["gw" + k], fr["gw" + k], smp, groups)
        vol = torch.cat((v_gwc, v_cat, smp.unsqueeze(1)), 1)
        c0 = _cf_dres(cx, vol, f"confidence0_{tag}", f"confidence1_{tag}")
        o1 = hourglass_pcw(cx, c0, f"confidence2_{tag}")
        o2 = hourglass_pcw(cx, o1, f"confidence3_{tag}")
        poss = F.softmax(_cf_classif(cx, o2, f"confidence_classif1_{tag}"), dim=1)
        return c0, o1, poss, smp

    g3, b3, g2, b2 = sd["gamma_s3"], sd["beta_s3"], sd["gamma_s2"], sd["beta_s2"]
    lo = up(cur - (g3 + 1) * var - b3, 2, H // 4, W // 4)
    hi = up(cur + (g3 + 1) * var + b3, 2, H // 4, W // 4)
    c0_s3, o1_s3, poss3, smp3 = stage("3", lo, hi, 14, 2, 40, "s3")
    pred1_s3 = torch.sum(poss3 * smp3, 1, keepdim=True)
    cur = pred1_s3.detach()
    var = torch.sum(poss3 * (cur - smp3) ** 2, 1, keepdim=True).sqrt()
    lo = up(cur - (g2 + 1) * var - b2, 2, H // 2, W // 2)
    hi = up(cur + (g2 + 1) * var + b2, 2, H // 2, W // 2)
    c0_s2, o1_s2, poss2, smp2 = stage("2", lo, hi, 10, 1, 20, "s2")
    pred1_s2 = torch.sum(poss2 * smp2, 1, keepdim=True)
    if not training:
        out = up(pred1_s2, 2, H, W).squeeze(1)
        return (out, cx) if return_ctx else out

    def head(x, p):
        c = F.conv3d(mish(convbn_3d(cx, x, p + ".0")), sd[p + ".2.weight"], None, 1, 1)
        return regression_head(c, maxdisp, H, W, align_corners=True)

    def sampled(x, p, smp, scale):
        pr = F.softmax(_cf_classif(cx, x, p), dim=1)
        return up(torch.sum(pr * smp, 1, keepdim=True), scale, H, W).squeeze(1)

    # evaluation order of the reference (cfnet.py:609-651): classif0, classif1, then the stage-3 and stage-2 side heads
    p0, p1 = head(cost0_4, "classif0"), head(out1_4, "classif1")
    out = [p0, p1, up(pred2_s4, 8, H, W).squeeze(1), sampled(c0_s3, "confidence_classif0_s3", smp3, 4),
           sampled(o1_s3, "confidence_classifmid_s3", smp3, 4), up(pred1_s3, 4, H, W).squeeze(1),
           sampled(c0_s2, "confidence_classif0_s2", smp2, 2), sampled(o1_s2, "confidence_classifmid_s2", smp2, 2),
           up(pred1_s2, 2, H, W).squeeze(1)]
    return (out, cx) if return_ctx else out
