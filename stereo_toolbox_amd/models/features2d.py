"""2-D feature CNNs feeding the cost-volume builders.

These are ON the critical path but their CONVOLUTIONS are not hand-written: they stay stock
PyTorch-ROCm (MIOpen) kernels.  Only their *structure* is dictated here -- by state-dict compatibility with the
reference checkpoints (reference models/GwcNet/gwcnet.py:12-65, models/PSMNet/submodule.py:57-132,
models/ACVNet/acv.py:15-54): module names, shapes and the Sequential indices must match so
`load_checkpoint_flexible` loads published weights.

Round 3: the glue AROUND the convolutions.  A rocprofv3 split of the extractor's train step (profiles/r03_feat2d_*.txt:
27.7 ms fwd+bwd for both 576x960 views) showed 45 % of it outside the convolutions: MIOpen BatchNorm 4.7 ms, NCHW<->NHWC
transposes around MIOpen's NHWC-only implicit-GEMM backward kernels 3.5 ms, stock elementwise adds / ReLU / counters
3.5 ms.  In train mode on a ROCm device the extractor therefore runs channels-last end to end (no transposes; the conv
weights are kept in channels_last memory format so that no per-call weight re-layout happens either) and every
`BatchNorm2d (+ReLU) (+residual add / + BatchNorm2d(downsample))` is ONE fused pass on the in-house channels-last BN
kernels of the 3-D path (stx_bn_stats -> stx_bn_finalize -> stx_bn_apply; backward stx_bn_bwd_reduce2 / _apply2),
through the same autograd function (ops.BnActFn).  The nn.BatchNorm2d / nn.ReLU modules stay as parameter containers.
Eval mode and CPU tensors keep the stock modules.  STX_FEAT2D_FUSED=0 switches the fused glue off (A/B).
"""
import math
import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..aggregation import _bn_state, _fold


def channels_last_weights_(module):
    """Store every Conv2d weight of `module` in channels_last memory format (values, shapes and state-dict keys are
    unchanged; `load_state_dict` / `copy_` preserve the format).  With channels-last activations MIOpen's NHWC kernels
    then take the weights as they are -- a contiguous weight would be re-laid-out by a small kernel in every forward,
    backward-data and backward-weight call (4 launches per convolution and step)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d) and m.weight.dim() == 4:
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return module


def fused_glue(x, *bns):
    """Whether the BatchNorm2d / ReLU / add glue runs on the in-house kernels for this call: fp32 on a ROCm device and
    either every BatchNorm involved in train mode (batch statistics, autograd) or -- inference -- every one in eval mode with autograd off (running statistics folded into one scale / shift pass per block)."""
    if os.environ.get("STX_FEAT2D_FUSED", "1") == "0":
        return False
    if x.dtype != torch.float32 or not ops.on_device(x):
        return False
    if ops.autocast_active():                 # --amp: the 2-D CNN is the caller's to run in fp16 / bf16 -- stock modules end to end
        return False
    if not all(isinstance(b, nn.modules.batchnorm._BatchNorm) for b in bns):
        return False
    if all(b.training for b in bns):
        return True
    return (not torch.is_grad_enabled() and not any(b.training for b in bns)
            and all(b.track_running_stats and b.running_mean is not None for b in bns))


class view_groups:
    """Context manager: the batch entering the extractor is `n` views stacked along the batch axis (left | right).  The
    convolutions run on the whole batch, every fused BatchNorm keeps per-view statistics and updates its running
    statistics view after view -- what the reference's separate extractor calls do (gwcnet.py:172-173) -- at half the
    convolution launches and twice the work per launch (the 64-channel layers at 1/4 resolution fill the chip only
    1.05 times at batch 1).  The count is per host thread (two threads running forwards side by side -- DataParallel
    replicas, an evaluation thread beside training -- must not see each other's grouping)."""
    _tls = threading.local()

    def __init__(self, n):
        self.new = int(n)

    @staticmethod
    def current():
        return getattr(view_groups._tls, "n", 1)

    def __enter__(self):
        self.old, view_groups._tls.n = view_groups.current(), self.new
        return self

    def __exit__(self, *exc):
        view_groups._tls.n = self.old
        return False


def _nhwc(t):
    """NCHW-logical tensor -> dense [B, H, W, C] (a view when `t` is channels_last already)."""
    v = t.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def cat_features(parts):
    """`torch.cat(parts, dim=1)` of NCHW-logical feature maps (reference gwcnet.py:59, acv.py:48).  On the device, for fp32
    channels_last maps: one coalesced pass through ops.cat_channels and, in backward, one pass that hands every branch a
    dense gradient (torch's cat falls back to a generic strided copy for channels_last operands, and its backward to narrow
    views that each cost a `contiguous()` copy downstream: 0.5 ms per train step at 576x960)."""
    parts = list(parts)
    if (os.environ.get("STX_FEAT2D_FUSED", "1") != "0" and os.environ.get("STX_FEAT2D_FUSED_CAT", "1") != "0" and 2 <= len(parts) <= 4
            and all(p.dtype == torch.float32 and ops.on_device(p) and p.shape[1] % 4 == 0
                    and p.is_contiguous(memory_format=torch.channels_last) for p in parts)):
        return ops.cat_channels([_nhwc(p) for p in parts]).permute(0, 3, 1, 2)
    return torch.cat(parts, dim=1)


def own_conv2d(x, conv):
    """Whether this convolution runs on the hand-written 2-D kernel (train mode, fp32, on the device; STX_FEAT2D_CONV=0: MIOpen)."""
    return (os.environ.get("STX_FEAT2D_CONV", "1") != "0" and x.dtype == torch.float32 and ops.on_device(x)
            and ops.conv2d_supported(conv) and x.shape[0] % view_groups.current() == 0)


def conv_bn_act(x, conv, bn, relu=False, residual=None, second=None):
    """act(BN(conv(x)) [+ residual | + BN2(z2)]): MIOpen convolution (channels-last), then for a train-mode BatchNorm2d one
    statistics pass and one fused normalise / add / ReLU pass, for an eval-mode one (inference) the fused pass alone.  x / residual: NCHW-logical; second = (z2, bn2) with z2 the raw
    output of the other branch's convolution.  Returns an NCHW-logical channels_last tensor."""
    G = view_groups.current()
    part = None
    if bn.training and own_conv2d(x, conv):
        # round 6: the 32 / 64-channel 3x3 stride-1 convolutions (39 of the GwcNet extractor's 55) on csrc/conv2d.hip, forward and
        # data gradient; the kernel's epilogue emits the BatchNorm statistics rows (no stx_bn_stats pass over z)
        zl, part = ops.Conv2dFn.apply(x, conv.weight, G)
        z = zl.permute(0, 3, 1, 2)
    else:
        z = F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)
        zl = _nhwc(z)
    if not bn.training:
        if torch.is_grad_enabled() and (z.requires_grad or bn.weight.requires_grad):
            raise ops.StxError("conv_bn_act: eval-mode BatchNorm under autograd has no fused backward -- gate the call "
                               "with fused_glue() over every BatchNorm of the block")
        # inference: running statistics folded into (scale, shift), cached per module like the 3-D path's
        sc, sh = _fold(bn)
        if second is not None:
            sc2, sh2 = _fold(second[1])
            y = ops.bn_apply(zl, sc, sh, _nhwc(second[0]), sc2, sh2, relu)
        else:
            y = ops.bn_apply(zl, sc, sh, None if residual is None else _nhwc(residual), None, None, relu)
        return y.permute(0, 3, 1, 2)
    st = _bn_state(bn, part if part is not None else ops.bn_stats(zl.detach(), G), zl.numel() // zl.shape[-1] // G, steps=G)
    if second is not None:
        z2l = _nhwc(second[0])
        bn2 = second[1]
        st2 = _bn_state(bn2, ops.bn_stats(z2l.detach(), G), z2l.numel() // z2l.shape[-1] // G, steps=G)
        y = ops.BnActFn.apply(zl, bn.weight, bn.bias, z2l, bn2.weight, bn2.bias, None, relu, st, st2, G)
    else:
        y = ops.BnActFn.apply(zl, bn.weight, bn.bias, None, None, None, None if residual is None else _nhwc(residual),
                              relu, st, None, G)
    return y.permute(0, 3, 1, 2)


def convbn_relu_chain(seq, x):
    """nn.Sequential of [convbn, ReLU]* optionally ending in a bare Conv2d (firstconv, lastconv, concatconv), fused."""
    i, n = 0, len(seq)
    while i < n:
        m = seq[i]
        if isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[0], nn.Conv2d):
            relu = i + 1 < n and isinstance(seq[i + 1], nn.ReLU)
            x = conv_bn_act(x, m[0], m[1], relu=relu)
            i += 2 if relu else 1
        else:
            x = m(x)
            i += 1
    return x


def _chain_bns(seq):
    """Every BatchNorm of a [convbn, ReLU]* chain: the fused path is taken only when ALL of them agree on the mode (a chain
    with one frozen BatchNorm under autograd must not take the inference branch of conv_bn_act, which has no backward)."""
    return [m[1] for m in seq if isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[0], nn.Conv2d)]


def run_head2d(seq, x):
    """`convbn + ReLU + Conv2d(1x1)` heads (GwcNet lastconv, ACVNet concatconv, PSMNet lastconv)."""
    if fused_glue(x, *_chain_bns(seq)):
        return convbn_relu_chain(seq, x)
    return seq(x)


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
        bns = [self.conv1[0][1], self.conv2[1]] + ([self.downsample[1]] if self.downsample is not None else [])
        if fused_glue(x, *bns):
            y = conv_bn_act(x, self.conv1[0][0], self.conv1[0][1], relu=True)
            if self.downsample is None:
                return conv_bn_act(y, self.conv2[0], self.conv2[1], residual=x)
            d = self.downsample[0]
            zd = F.conv2d(x, d.weight, d.bias, d.stride, d.padding, d.dilation, d.groups)
            return conv_bn_act(y, self.conv2[0], self.conv2[1], second=(zd, self.downsample[1]))
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
        if fused_glue(x, *_chain_bns(self.firstconv)):
            x = convbn_relu_chain(self.firstconv, x.contiguous(memory_format=torch.channels_last))
        else:
            x = self.firstconv(x)
        x = self.layer1(x)
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


class _SplitViewsFn(torch.autograd.Function):
    """(both[:B], both[B:]) of a batch-stacked two-view map.  Plain slicing is the same forward, but its backward builds each
    half's gradient as `zeros(both.shape)` in NCHW memory + a copy of the channels-last half into it (a strided copy: 0.1 ms
    per 320 x 144 x 240 half) and hands the extractor an NCHW gradient that the next channels-last consumer re-lays once
    more (0.15-0.23 ms); here the two halves are concatenated once, in their own memory format."""

    @staticmethod
    def forward(ctx, both, B):
        ctx.B, ctx.n, ctx.fmt = B, both.shape[0], (torch.channels_last if both.dim() == 4 and
                                                   both.is_contiguous(memory_format=torch.channels_last) else torch.contiguous_format)
        return both[:B], both[B:]

    @staticmethod
    def backward(ctx, gl, gr):
        if gl is None and gr is None:
            return None, None
        ref = gl if gl is not None else gr
        if gl is None:
            gl = ref.new_zeros((ctx.B,) + tuple(ref.shape[1:]))
        if gr is None:
            gr = ref.new_zeros((ctx.n - ctx.B,) + tuple(ref.shape[1:]))
        return torch.cat((gl.contiguous(memory_format=ctx.fmt), gr.contiguous(memory_format=ctx.fmt)), 0), None


def _split_views(both, B):
    if isinstance(both, dict):
        halves = {k: _SplitViewsFn.apply(v, B) for k, v in both.items()}
        return {k: h[0] for k, h in halves.items()}, {k: h[1] for k, h in halves.items()}
    return _SplitViewsFn.apply(both, B)


def run_pair(extractor, left, right, training):
    """Run a 2-D extractor on both views.  In eval mode the two views share one batched pass (same
    numbers, half the launches); in train mode they stay separate calls so that BatchNorm batch
    statistics and running-stat updates match the reference exactly (gwcnet.py:172-173)."""
    if training:
        bns = [m for m in extractor.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        if (os.environ.get("STX_FEAT2D_PAIRED", "1") != "0" and left.shape == right.shape and bns and fused_glue(left, *bns)
                and all(b.momentum is not None and not isinstance(b, nn.SyncBatchNorm) for b in bns)
                and getattr(extractor, "fused_everywhere", False)):
            # one batched pass, per-view BatchNorm statistics (see view_groups); only for extractors whose EVERY BatchNorm
            # goes through the fused glue (a stock BatchNorm module would pool the two views)
            with view_groups(2):
                both = extractor(torch.cat((left, right), 0))
            return _split_views(both, left.shape[0])
        return extractor(left), extractor(right)
    both = extractor(torch.cat((left, right), 0))
    B = left.shape[0]
    if isinstance(both, dict):
        return {k: v[:B] for k, v in both.items()}, {k: v[B:] for k, v in both.items()}
    return both[:B], both[B:]
