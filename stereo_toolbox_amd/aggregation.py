"""Host-side orchestration of the Conv3d aggregation: maps the reference's module structure
(`convbn_3d` = nn.Sequential(Conv3d, BatchNorm3d), ConvTranspose3d + BatchNorm3d, bare Conv3d) onto
the fused HIP kernels.

The nn.Conv3d / nn.BatchNorm3d objects are kept as *parameter containers* so that state-dict keys,
DDP gradient hooks, SyncBatchNorm conversion and optimizers see exactly the reference's modules
(SURVEY.md 8b); their own forward() is never called on the hot path.

Three execution modes of one block  y = act(BN(conv(x)) [+ BN2(conv2(x2))] [+ residual]):
  * inference (no grad, eval BN): BN folded into the conv epilogue, one kernel per conv;
  * training  (train BN): conv emits raw z + batch-stat partials -> bn_finalize -> bn_apply;
  * eval BN with autograd: as training but with the running statistics.
"""
import os
import threading

import torch
import torch.nn as nn

from . import ops


def _is_transposed(conv):
    return isinstance(conv, nn.ConvTranspose3d)


def _conv_cfg(conv):
    ks = conv.kernel_size[0]
    stride = conv.stride[0]
    if _is_transposed(conv):
        k3 = ks == 3 and stride == 2 and conv.padding[0] == 1 and conv.output_padding[0] == 1      # GwcNet / PSMNet family
        k4 = ks == 4 and stride == 2 and conv.padding[0] == 1 and conv.output_padding[0] == 0      # IGEV family (out = 2 in)
        if not (k3 or k4) or len(set(conv.kernel_size)) != 1 or len(set(conv.stride)) != 1:
            raise ops.StxError("only ConvTranspose3d(k=3, s=2, p=1, op=1) and ConvTranspose3d(k=4, s=2, p=1) are on the hot path")
    else:
        if not ((ks == 3 and conv.padding[0] == 1 and stride in (1, 2)) or (ks == 1 and conv.padding[0] == 0 and stride == 1)):
            raise ops.StxError(f"unsupported Conv3d k={ks} s={stride} p={conv.padding}")
    if conv.bias is not None or conv.groups != 1:
        raise ops.StxError("hot-path convolutions are bias-free and dense")
    return ks, stride


def _fold(bn):
    """Eval-mode BN as per-channel (scale, bias); cached on the module, keyed by tensor versions and storage addresses
    (in-place edits through `.data` move neither: ops.invalidate_caches / ops.set_weight_cache).  A train() <-> eval()
    transition of the module drops the entry."""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr())
    hit = bn.__dict__.get("_stx_fold") if ops._CACHE_ENABLED else None
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    with torch.no_grad():
        scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
        bias = (bn.bias - bn.running_mean * scale).contiguous()
    if ops._CACHE_ENABLED:
        bn.__dict__["_stx_fold"] = (key, scale, bias)
    return scale, bias


def _needs_grad(*tensors_and_modules):
    if not torch.is_grad_enabled():
        return False
    for t in tensors_and_modules:
        if t is None:
            continue
        if isinstance(t, torch.Tensor):
            if t.requires_grad:
                return True
        else:
            if any(p.requires_grad for p in t.parameters()):
                return True
    return False


def _infer_conv(x, conv, scale, bias, residual, relu):
    ks, stride = _conv_cfg(conv)
    if _is_transposed(conv) and ks == 4:
        if residual is not None:
            raise ops.StxError("ConvTranspose3d(k4) with a fused residual is not wired (no model uses it)")
        return ops.deconv4_forward(x, conv.weight.detach(), scale, bias, int(relu), owner=conv.weight)
    if _is_transposed(conv):
        wp = ops.pack_weight(conv.weight.detach(), 2, owner=conv.weight)
        return ops.deconv3d_forward(x, wp, conv.weight.shape[1], scale=scale, bias=bias, residual=residual, relu=relu)[0]
    if ops._is_c1(conv.weight, ks, stride, False) and scale is None and bias is None and not relu:
        return ops.conv3d_c1_forward(x, conv.weight.detach().contiguous(), residual)
    if x.shape[-1] != conv.weight.shape[1]:     # a volume that already carries its zero pad channels (ops.sampled_volume)
        if x.shape[-1] != (conv.weight.shape[1] + 7) // 8 * 8:
            raise ops.StxError(f"conv: input has {x.shape[-1]} channels, weight expects {conv.weight.shape[1]}")
        wpad = ops.pad_weight_channels(conv.weight.detach(), x.shape[-1])
        return ops.conv3d_forward(x, ops.pack_weight(wpad, 0), conv.weight.shape[0], ks, stride, scale, bias, residual, relu)[0]
    if x.shape[-1] % 8 != 0:       # odd input widths (CFNet cascade volumes): zero-padded GEMM-K, packed copy not cached
        xp, wpad = ops.pad_input_channels(x, conv.weight.detach())
        return ops.conv3d_forward(xp, ops.pack_weight(wpad, 0), conv.weight.shape[0], ks, stride, scale, bias, residual, relu)[0]
    wp = ops.pack_weight(conv.weight.detach(), 0, owner=conv.weight)
    return ops.conv3d_forward(x, wp, conv.weight.shape[0], ks, stride, scale, bias, residual, relu)[0]


class deferred_bn_counters:
    """Context manager for a model forward: the `num_batches_tracked += 1` of every train-mode BatchNorm inside is
    collected and applied as ONE multi-tensor add on exit instead of one 5-us kernel per layer (GwcNet_GC: 26 3-D and
    112 2-D BatchNorm calls per train step = 0.6 ms of serialized launches).  Re-entrant; modules with momentum=None
    (cumulative average: the factor needs the counter's value) keep the immediate update.  State is per host thread."""
    _tls = threading.local()

    @staticmethod
    def _state():
        st = deferred_bn_counters._tls
        if not hasattr(st, "active"):
            st.active, st.pending = 0, []
        return st

    def __enter__(self):
        self._state().active += 1
        ops.bn_defer_reset_if_stale()        # a model forward outside any backward pass: nothing can be legitimately outstanding
        return self

    def __exit__(self, *exc):
        st = self._state()
        st.active -= 1
        if st.active == 0 and st.pending:
            pend, st.pending = st.pending, []
            by_dev = {}
            for t in pend:        # a module that ran twice (left and right view) counts twice: one entry, increment 2
                d = by_dev.setdefault(t.device, {})
                ent = d.setdefault(id(t), [t, 0])
                ent[1] += 1
            for d in by_dev.values():
                torch._foreach_add_([e[0] for e in d.values()], [e[1] for e in d.values()])
        return False


def _bn_state(bn, partials, count, steps=1):
    """`steps`: statistic updates this call stands for (the grouped 2-D path: one per view)."""
    training = bn.training
    sync = None
    if training and isinstance(bn, nn.SyncBatchNorm):
        # trainer/trainer_torchrun.py:112-113 (`SyncBatchNorm.convert_sync_batchnorm`): statistics over all replicas
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(bn.process_group) > 1:
            sync = (bn.process_group, dist.get_world_size(bn.process_group))
    if training:
        bn.__dict__.pop("_stx_fold", None)          # the running statistics are about to move
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        st = deferred_bn_counters._state()
        if st.active and bn.momentum is not None:
            st.pending.extend([bn.num_batches_tracked] * steps)
        else:
            bn.num_batches_tracked.add_(steps)
    if bn.momentum is not None:
        momentum = bn.momentum
    elif training and bn.track_running_stats and bn.num_batches_tracked is not None:
        # momentum=None is torch's cumulative moving average: factor 1 / num_batches_tracked (after the increment);
        # one host read per layer and step -- only for modules built that way (the reference's models never are)
        momentum = 1.0 / float(bn.num_batches_tracked.item())
    else:
        momentum = 0.0
    return {"training": training, "partials": partials, "count": count, "running_mean": bn.running_mean,
            "running_var": bn.running_var, "momentum": momentum, "eps": bn.eps, "sync": sync}


def _raw_conv(x, conv, want):
    """Raw convolution output + BatchNorm partial rows through autograd (ConvRawFn; ConvTranspose3d(k4) through its
    parity-class form)."""
    ks, stride = _conv_cfg(conv)
    if _is_transposed(conv) and ks == 4:
        return ops.deconv4_raw(x, conv.weight, want)
    return ops.ConvRawFn.apply(x, conv.weight, ks, stride, _is_transposed(conv), want)


@ops.fp32_region
def shared_input_convs(x, blocks):
    """Train-mode raw outputs of several `convbn_3d` blocks (nn.Sequential(Conv3d, BatchNorm3d)) reading the SAME activation
    `x`, through one autograd node whose backward accumulates the input gradients inside the producing kernels instead of
    in volume-sized `add` launches (ops.SharedInputConvsFn).  Returns one `raw` triple per block for conv_block's `raw=` /
    `second_raw=` arguments, or None when the combination is not served (transposed / odd widths / inference): the caller then
    takes the block-by-block path."""
    cfgs, ws = [], []
    for seq in blocks:
        conv, bn = seq[0], seq[1]
        if _is_transposed(conv) or not bn.training or conv.weight.shape[1] != x.shape[-1] or conv.weight.shape[1] % 8 \
                or conv.weight.shape[0] % 8:
            return None
        ks, stride = _conv_cfg(conv)
        cfgs.append((ks, stride, True))
        ws.append(conv.weight)
    if not _needs_grad(x, *[m for seq in blocks for m in (seq[0], seq[1])]):
        return None
    outs = ops.SharedInputConvsFn.apply(x, tuple(cfgs), *ws)
    return [(outs[2 * k], outs[2 * k + 1]) for k in range(len(blocks))]


@ops.fp32_region
def conv_block(x, conv, bn=None, relu=False, second=None, residual=None, mish=False, leaky=False, raw=None, second_raw=None):
    """One fused block on NDHWC tensors.
    leaky: LeakyReLU(0.01) instead of ReLU (IGEV family, activation code 3 of the kernels).
    second = (x2, conv2, bn2): adds BN2(conv2(x2)) before the activation (hourglass redir path).
    residual: NDHWC tensor added before the activation.
    mish: Mish instead of ReLU (PCWNet / CFNet family): fused into the conv epilogue in inference and into the BatchNorm
    apply / backward passes in training (differentiated at the recomputed pre-activation value).
    raw / second_raw = (z, partial rows) from shared_input_convs: the (train-mode) raw output of `conv` / of second's conv
    computed earlier; `x` / second[0] are then not read."""
    if (mish and relu) or (leaky and (relu or mish)):
        raise ops.StxError("conv_block: relu, mish and leaky are mutually exclusive")
    if second is not None and residual is not None:
        raise ops.StxError("conv_block: `second` and `residual` are mutually exclusive")
    mods = [conv, bn] + ([second[1], second[2]] if second is not None else [])
    grad = (raw is not None or second_raw is not None
            or _needs_grad(x, residual, *(m for m in mods if m is not None), *([second[0]] if second else [])))
    train_bn = (bn is not None and bn.training) or (second is not None and second[2].training)
    if mish and (grad or train_bn) and (bn is None or residual is not None):
        # autograd path without a BatchNorm to fuse into: Mish as its own streaming pass (forward + backward kernels)
        return ops.mish(conv_block(x, conv, bn, relu=False, second=second, residual=residual))
    if mish:
        # activation code 2: conv epilogue (inference) / BN apply and the two BN backward passes (train), no extra pass
        relu = 2
    if leaky:
        relu = 3

    if not grad and not train_bn:   # ---- inference: everything folded into conv epilogues
        if second is not None:
            s2, b2 = _fold(second[2])
            residual = _infer_conv(second[0], second[1], s2, b2, None, False)
        if bn is not None:
            s1, b1 = _fold(bn)
        else:
            s1 = b1 = None
        return _infer_conv(x, conv, s1, b1, residual, relu)

    # ---- autograd path
    want = bn is not None and bn.training
    z1, part1 = raw if raw is not None else _raw_conv(x, conv, want)
    if bn is None:
        y = z1
        if residual is not None:
            y = y + residual
        if relu == 3:
            return torch.nn.functional.leaky_relu(y, 0.01)
        return torch.relu(y) if relu else y
    count1 = z1.numel() // z1.shape[-1]
    st1 = _bn_state(bn, part1 if want else None, count1)
    if second is not None:
        x2, conv2, bn2 = second
        want2 = bn2.training
        z2, part2 = second_raw if second_raw is not None else _raw_conv(x2, conv2, want2)
        st2 = _bn_state(bn2, part2 if want2 else None, z2.numel() // z2.shape[-1])
        return ops.BnActFn.apply(z1, bn.weight, bn.bias, z2, bn2.weight, bn2.bias, None, relu, st1, st2)
    # a 3x3x3 stride-1 block without residual: its BatchNorm's backward-apply pass rides inside the convolution's march
    # weight-gradient kernel (ops._PendingBn; the convolution's backward node falls back to the stand-alone pass otherwise)
    defer = (residual is None and want and relu in (0, 1, False, True) and not _is_transposed(conv) and _conv_cfg(conv) == (3, 1)
             and conv.weight.shape[0] % 32 == 0 and conv.weight.shape[1] % 32 == 0 and not st1.get("sync")
             and os.environ.get("STX_BN_BWD_IN_WGRAD", "1") != "0")
    if defer:
        ops.bn_defer_reset_if_stale()
    return ops.BnActFn.apply(z1, bn.weight, bn.bias, None, None, None, residual, relu, st1, None, 1, defer)


@ops.fp32_region
def convbn_block(x, seq, relu=False, second=None, residual=None, mish=False, raw=None, second_raw=None):
    """`seq` = nn.Sequential(conv, bn) as built by convbn_3d (or (ConvTranspose3d, BatchNorm3d))."""
    sec = None
    if second is not None:
        sec = (second[0], second[1][0], second[1][1])
    return conv_block(x, seq[0], seq[1], relu=relu, second=sec, residual=residual, mish=mish, raw=raw, second_raw=second_raw)
