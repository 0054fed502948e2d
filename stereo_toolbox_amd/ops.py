"""Tensor-level host side of the HIP kernels: argument checking, workspaces, autograd.

Everything here launches hand-written gfx950 kernels through the C-ABI (include/stx_hip.h) on the
caller's current HIP stream.  There is no CPU or stock-torch fallback for these ops: tensors that
are not fp32 on a ROCm device raise, and a missing library raises at first use.

Activation convention inside the 3-D path: dense channels-last tensors of shape [B, D, H, W, C]
("NDHWC").  `to_ncdhw` / `to_ndhwc` give zero-copy logical views for the public API.
"""
import ctypes

import torch

from ._capi import StxError, get_lib

_P = ctypes.c_void_p


# --------------------------------------------------------------------------------------- plumbing
class _DevPtr(ctypes.c_void_p):
    """A device pointer that remembers the device of the tensor it came from: _call() reads the launch device off its
    own arguments, so nothing survives between launches (an exception between building the arguments and the launch
    cannot leak state into the next one)."""


def _stream(dev=None):
    return _P(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    if t is None:
        return None
    p = _DevPtr(t.data_ptr())
    p.dev = t.device
    return p


def on_device(t):
    """Whether `t` lives where the kernels of this library run (a ROCm device)."""
    return t.is_cuda


def _chk(t, name, dims=None):
    if t is None:
        return
    if not t.is_cuda:
        raise StxError(f"{name}: expected a ROCm device tensor, got {t.device} -- the cost-volume hot path has "
                       "no CPU fallback (the CPU oracle lives in oracle/ and is test-only)")
    if t.dtype != torch.float32:
        raise StxError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise StxError(f"{name}: expected a dense contiguous tensor, strides {t.stride()}")
    if dims is not None and t.dim() != dims:
        raise StxError(f"{name}: expected {dims} dims, got shape {tuple(t.shape)}")


def _call(name, *args):
    """Launch one C-ABI entry point on the device that OWNS the operands, on that device's current stream
    (`model.to('cuda:1')` without `torch.cuda.set_device(1)` -- the way evaluation/sceneflow_test.py:22 and
    generalization_eval.py use `device=` -- must not launch on cuda:0's stream).  Operands on different devices raise."""
    dev = None
    for a in args:
        if isinstance(a, _DevPtr):
            if dev is None:
                dev = a.dev
            elif a.dev != dev:
                raise StxError(f"{name}: operands live on different devices ({dev} and {a.dev})")
    if dev is not None and dev.type == "cuda" and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            get_lib().call(name, *args, _stream(dev))
    else:
        get_lib().call(name, *args, _stream(dev))


# ------------------------------------------------------------------------- mixed-precision callers
_LOW = (torch.float16, torch.bfloat16)


def _to_fp32(v):
    if isinstance(v, torch.Tensor):
        return v.float() if v.dtype in _LOW else v          # a differentiable cast: the gradient goes back in the caller's dtype
    if isinstance(v, dict):
        return {k: _to_fp32(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(_to_fp32(x) for x in v)
    return v


def autocast_active():
    return torch.is_autocast_enabled("cuda") or torch.is_autocast_enabled("cpu")


def fp32_region(fn):
    """The hot path computes in fp32 (north_star; the reference's own FoundationStereo builds its volume outside autocast for
    the same reason, FoundationStereo/submodule.py:388-397).  Under `torch.amp.autocast` -- the reference Trainer's `--amp`,
    trainer/trainer_torchrun.py:219,274,286-294 -- the stock 2-D CNN hands over fp16 / bf16 feature maps: a decorated entry
    point casts its low-precision tensor arguments (also inside dicts / lists) to fp32 and runs with autocast switched off,
    like the operators on autocast's own fp32 list; outputs are fp32.  The casts are ordinary autograd nodes, so a
    `GradScaler`-scaled loss back-propagates through the fp32 kernels and reaches the 2-D CNN in its own dtype.  Outside
    autocast nothing changes: a low-precision tensor is still refused by `_chk` (there is no fp16 kernel to fall back to)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        if not autocast_active():
            return fn(*args, **kwargs)
        with torch.autocast("cuda", enabled=False), torch.autocast("cpu", enabled=False):
            return fn(*_to_fp32(args), **_to_fp32(kwargs))
    return wrapped


def to_ncdhw(x):
    """[B,D,H,W,C] dense -> logical [B,C,D,H,W] view (channels_last_3d strides, no copy)."""
    return x.permute(0, 4, 1, 2, 3)


def to_ndhwc(x):
    """logical [B,C,D,H,W] (any strides) -> dense [B,D,H,W,C]."""
    return x.permute(0, 2, 3, 4, 1).contiguous()


class _Workspace:
    """Grow-only scratch buffers per (device, stream, tag): wgrad slabs, BN partial sums."""

    def __init__(self):
        self.bufs = {}

    def get(self, tag, nfloats, device):
        # keyed by the launching stream as well: two streams (or autograd threads on different streams) must not
        # share a slab whose contents live from one launch to the next
        sid = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
        key = (device.index, sid, tag)
        b = self.bufs.get(key)
        if b is None or b.numel() < nfloats:
            b = torch.empty(int(nfloats), dtype=torch.float32, device=device)
            self.bufs[key] = b
        return b


_WS = _Workspace()


_CACHE_ENABLED = True


def set_weight_cache(enabled):
    """Turn the inference-time caches (packed conv weights, folded eval-mode BatchNorm) on or off.  They are keyed by
    the owner tensors' autograd version counters and storage addresses, which every regular update moves (optimizer
    steps, `load_state_dict`, `copy_`, in-place ops on the parameter).  An in-place edit THROUGH `.data`
    (`p.data.mul_()`, `bn.running_var.data.fill_()`: EMA / weight-surgery code) does not move them: call
    `invalidate_caches(model)` after such edits, or switch the caches off here while doing them."""
    global _CACHE_ENABLED
    _CACHE_ENABLED = bool(enabled)


def invalidate_caches(model):
    """Drop every cached packed weight / folded BatchNorm of `model` (see set_weight_cache)."""
    for m in model.modules():
        m.__dict__.pop("_stx_fold", None)
    for p in model.parameters():
        p.__dict__.pop("_stx_packed", None)


def pack_weight(w, mode, owner=None):
    """Device-side re-layout of a torch conv weight [A][B][k,k,k] into MFMA B-operand order.
    mode 0: conv fwd / deconv dgrad;  1: stride-1 conv dgrad;  2: deconv fwd / stride-2 conv dgrad.
    `owner` (the nn.Parameter) enables caching for inference: the packed copy is stored ON the
    parameter object (so it dies with it -- a dict keyed by data_ptr would hand stale weights to a
    new model whose storage reuses the address) and is refreshed when the version counter moves
    (`.data` edits do not move it: see set_weight_cache / invalidate_caches)."""
    _chk(w, "weight", 5)
    if owner is not None and _CACHE_ENABLED:
        cache = owner.__dict__.setdefault("_stx_packed", {})
        hit = cache.get(mode)
        if hit is not None and hit[0] == (owner._version, owner.data_ptr()):
            return hit[1]
    A, Bd = w.shape[0], w.shape[1]
    T = w.shape[2] * w.shape[3] * w.shape[4]
    K, N = (Bd, A) if mode == 0 else (A, Bd)
    n = get_lib().raw("stx_conv3d_packed_floats")(K, N, T)
    wp = torch.empty(n, dtype=torch.float32, device=w.device)
    _call("stx_conv3d_pack_weight", _p(w), _p(wp), A, Bd, T, mode)
    if owner is not None and _CACHE_ENABLED:
        cache[mode] = ((owner._version, owner.data_ptr()), wp)
    return wp


def _pad_channels(t, C):
    """Zero-pad the last (channel) axis of an NDHWC tensor / dim 0 or 1 of a weight."""
    if t.shape[-1] == C:
        return t
    out = t.new_zeros(*t.shape[:-1], C)
    out[..., : t.shape[-1]] = t
    return out


def pad_input_channels(x, w):
    """Zero-pad the channel axis of an NDHWC activation and dim 1 of a conv weight [Cout, Cin, k, k, k] to the next
    multiple of 8 (the implicit-GEMM kernels take K in steps of 8)."""
    Cin = x.shape[-1]
    Cp = (Cin + 7) // 8 * 8
    if Cp == Cin:
        return x, w
    wp = w.new_zeros(w.shape[0], Cp, *w.shape[2:])
    wp[:, :Cin] = w
    return _pad_channels(x, Cp), wp


def pad_weight_channels(w, Cp):
    """Zero-pad dim 1 (input channels) of a conv weight [Cout, Cin, k, k, k] to Cp."""
    if w.shape[1] == Cp:
        return w
    wp = w.new_zeros(w.shape[0], Cp, *w.shape[2:])
    wp[:, : w.shape[1]] = w
    return wp


# --------------------------------------------------------------------------------------- raw convs
def conv_out_dims(D, H, W, ks, stride):
    pad = ks // 2
    return tuple((d + 2 * pad - ks) // stride + 1 for d in (D, H, W))


def conv3d_forward(x, wp, Cout, ks, stride, scale=None, bias=None, residual=None, relu=False, want_stats=False):
    """out = act(conv(x) * scale + bias + residual); optional BN partial sums of the raw output."""
    _chk(x, "x", 5)
    B, D, H, W, Cin = x.shape
    Do, Ho, Wo = conv_out_dims(D, H, W, ks, stride)
    out = torch.empty(B, Do, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        rows = get_lib().raw("stx_conv3d_fwd_stat_rows")(B, D, H, W, Cin, Cout, ks, stride)   # (exactly the rows the launch writes)
        stats = torch.empty(rows, 2, Cout, dtype=torch.float32, device=x.device)
    _call("stx_conv3d_fwd", _p(x), _p(wp), _p(out), _p(scale), _p(bias), _p(residual), _p(stats), B, D, H, W, Cin,
          Cout, ks, stride, int(relu))
    return out, stats


def deconv3d_forward(x, wp, Cout, out_dims=None, scale=None, bias=None, residual=None, relu=False, want_stats=False):
    """ConvTranspose3d(k3, s2, p1, op1) (out = 2*in), same fused epilogue as conv3d_forward."""
    _chk(x, "x", 5)
    B, D, H, W, Cin = x.shape
    Do, Ho, Wo = out_dims if out_dims is not None else (2 * D, 2 * H, 2 * W)
    if Cout > 64:
        # The transposed-conv kernel keeps 4 parity-class accumulators per 32-column block and is instantiated for
        # N <= 64.  Wider outputs (PCWNet: ConvTranspose3d(128, 128), dgrad of Conv3d(128, 128, s2)) run as 64-channel
        # slices of the packed weights [tap][K/8][N/32][256] and are concatenated (small 1/16-1/32-resolution volumes).
        if Cout % 64 != 0:
            raise StxError(f"deconv3d_forward: Cout={Cout} > 64 must be a multiple of 64")
        nt = Cout // 32
        wv = wp.view(-1, Cin // 8, nt, 256)
        outs, sts = [], []
        for h in range(Cout // 64):
            sl = slice(64 * h, 64 * h + 64)
            o, st_ = deconv3d_forward(x, wv[:, :, 2 * h:2 * h + 2].contiguous(), 64, (Do, Ho, Wo),
                                      None if scale is None else scale[sl].contiguous(),
                                      None if bias is None else bias[sl].contiguous(),
                                      None if residual is None else residual[..., sl].contiguous(), relu, want_stats)
            outs.append(o)
            sts.append(st_)
        return torch.cat(outs, -1), (torch.cat(sts, -1) if want_stats else None)
    out = torch.empty(B, Do, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        nb = get_lib().raw("stx_deconv3d_fwd_blocks")(D, H, W)
        stats = torch.empty(B * nb, 2, Cout, dtype=torch.float32, device=x.device)
    _call("stx_deconv3d_fwd", _p(x), _p(wp), _p(out), _p(scale), _p(bias), _p(residual), _p(stats), B, D, H, W, Cin,
          Cout, Do, Ho, Wo, int(relu))
    return out, stats


def conv3d_wgrad(fine, coarse, ks, stride):
    """G[cc][cf][tap] = sum_o fine[S*o+tap-pad][cf] * coarse[o][cc] (see conv3d.hip)."""
    B, Df, Hf, Wf, CF = fine.shape
    _, Dc, Hc, Wc, CC = coarse.shape
    n = get_lib().raw("stx_conv3d_wgrad_workspace_floats")(B, Dc, Hc, Wc, CF, CC, ks, stride)
    ws = _WS.get("wgrad", n, fine.device)
    dw = torch.empty(CC, CF, ks ** 3, dtype=torch.float32, device=fine.device)
    _call("stx_conv3d_wgrad", _p(fine), _p(coarse), _p(dw), _p(ws), B, Df, Hf, Wf, CF, Dc, Hc, Wc, CC, ks, stride)
    return dw


def conv3d_wgrad_bn(x, gy, pend):
    """Weight gradient of a 3x3x3 stride-1 convolution from the gradient BEHIND its train-mode BatchNorm (+ ReLU): returns
    (dw [CC, CF, 27], dz) -- see stx_conv3d_wgrad_bn.  `pend` = the record BnActFn.backward left (_PendingBn)."""
    B, D, H, W, CF = x.shape
    CC = gy.shape[-1]
    n = get_lib().raw("stx_conv3d_wgrad_workspace_floats")(B, D, H, W, CF, CC, 3, 1)
    ws = _WS.get("wgrad", n, x.device)
    dw = torch.empty(CC, CF, 27, dtype=torch.float32, device=x.device)
    dz = torch.empty_like(gy)
    _call("stx_conv3d_wgrad_bn", _p(x), _p(gy), _p(pend.z), _p(pend.scale), _p(pend.shift), _p(pend.mean), _p(pend.invstd),
          _p(pend.gamma), _p(pend.sums), float(pend.inv_n), int(pend.act), _p(dz), _p(dw), _p(ws), B, D, H, W, CF, CC)
    return dw, dz


class _PendingBn:
    """What BnActFn.backward hands to the backward of the convolution that produced its input when the BatchNorm's
    `bn_bwd_apply` pass is folded into that convolution's weight-gradient kernel (conv_block decides: `defer`): the
    operands of dz = gamma invstd (gy' - sum_g / n - xhat sum_gx / n).  Attached to the gradient tensor it describes."""
    __slots__ = ("g", "z", "scale", "shift", "mean", "invstd", "gamma", "sums", "inv_n", "act")

    def materialize(self):
        """The BatchNorm-backward apply pass on its own (a consumer that cannot fold it: frozen weights, other kernels)."""
        C = self.z.shape[-1]
        nvox = self.z.numel() // C
        dz = torch.empty_like(self.z)
        _call("stx_bn_bwd_apply2", _p(self.g), None, _p(self.z), _p(self.mean), _p(self.invstd), _p(self.gamma), None, None,
              None, None, _p(self.scale), _p(self.shift), None, None, _p(self.sums), _p(dz), None, None, nvox, C, int(self.act), 1)
        return dz


# Safety net of the deferral: a record that no convolution backward consumed means that convolution differentiated the
# UN-normalised gradient.  Every backward pass that defers queues one end-of-pass callback that raises in that case.
_BN_DEFER = {"outstanding": 0, "armed": False}


def _bn_defer_check():
    n, _BN_DEFER["outstanding"], _BN_DEFER["armed"] = _BN_DEFER["outstanding"], 0, False
    if n:
        raise StxError(f"{n} deferred BatchNorm-backward record(s) were not consumed by their convolution's backward node: "
                       "the gradients of this backward pass are wrong (set STX_BN_BWD_IN_WGRAD=0 and report)")


def _defer_pending_bn(g, pend):
    g._stx_pending_bn = pend
    _BN_DEFER["outstanding"] += 1
    if not _BN_DEFER["armed"]:
        _BN_DEFER["armed"] = True
        torch.autograd.Variable._execution_engine.queue_callback(_bn_defer_check)


def bn_defer_reset_if_stale():
    """Called from the FORWARD side of a deferring block (aggregation.conv_block): a backward pass that died with an exception
    after a deferral (out of memory, a user hook) never ran its end-of-pass callback and left the counters armed -- the
    safety net would stay off for the rest of the process.  Outside a backward pass nothing can be legitimately
    outstanding, so the state is reset."""
    if not _BN_DEFER["armed"]:
        return
    task_id = getattr(torch._C, "_current_graph_task_id", None)       # (a private accessor: a torch build without it skips the reset
    if task_id is None or task_id() == -1:                           #  outside-backward check and resets whenever a forward starts)
        _BN_DEFER["outstanding"], _BN_DEFER["armed"] = 0, False


def _take_pending_bn(g):
    """The deferred-BatchNorm record BnActFn.backward attached to gradient tensor `g`, if any (it travels ON the tensor object:
    no global table to leak when a backward pass is interrupted, nothing shared between threads)."""
    pend = getattr(g, "_stx_pending_bn", None) if g is not None else None
    if pend is not None:
        del g._stx_pending_bn
        _BN_DEFER["outstanding"] -= 1
        if pend.g is not g:                        # (cannot happen; a stale record must never be applied to another tensor)
            raise StxError("deferred BatchNorm backward: record attached to a different gradient tensor")
    return pend


def _is_c1(w, ks, stride, transposed):
    """Classifier tail Conv3d(Cin, 1, 3, padding=1): served by the VALU kernels of conv_c1.hip."""
    return (not transposed) and ks == 3 and stride == 1 and w.shape[0] == 1 and w.shape[1] % 16 == 0 and w.shape[1] <= 64


def conv3d_c1_forward(x, w, residual=None):
    """x [B,D,H,W,Cin], w [1,Cin,3,3,3] -> [B,D,H,W,1] (+ residual [B,D,H,W,1])."""
    _chk(x, "x", 5)
    _chk(w, "weight", 5)
    B, D, H, W, Cin = x.shape
    out = torch.empty(B, D, H, W, 1, dtype=torch.float32, device=x.device)
    _call("stx_conv3d_c1_fwd", _p(x), _p(w), _p(residual), _p(out), B, D, H, W, Cin)
    return out


def conv3d_c1_wgrad(x, gy):
    B, D, H, W, Cin = x.shape
    ws = _WS.get("c1wgrad", get_lib().raw("stx_conv3d_c1_wgrad_workspace_floats")(Cin), x.device)
    dw = torch.empty(1, Cin, 3, 3, 3, dtype=torch.float32, device=x.device)
    _call("stx_conv3d_c1_wgrad", _p(x), _p(gy), _p(dw), _p(ws), B, D, H, W, Cin)
    return dw


def conv3d_c1_dgrad(gy, w, Cin):
    """gy [B,D,H,W,1], w [1,Cin,3,3,3] -> gx [B,D,H,W,Cin]."""
    B, D, H, W, _ = gy.shape
    gx = torch.empty(B, D, H, W, Cin, dtype=torch.float32, device=gy.device)
    _call("stx_conv3d_c1_dgrad", _p(gy), _p(w), _p(gx), B, D, H, W, Cin)
    return gx


class ConvRawFn(torch.autograd.Function):
    """z = conv(x, w) (or transposed conv), raw output + BN partial sums; backward = dgrad + wgrad
    on the same MFMA kernels with re-packed weights."""

    @staticmethod
    def forward(ctx, x, w, ks, stride, transposed, want_stats):
        _chk(x, "x", 5)
        ctx.cin_true = None
        ctx.x_prepadded = False
        if not transposed and x.shape[-1] != w.shape[1]:
            # an activation that already carries zero pad channels up to the GEMM-K step (ops.sampled_volume): pad the
            # weight only; the pad channels' input gradient is returned (and ignored by the producer)
            if x.shape[-1] != (w.shape[1] + 7) // 8 * 8:
                raise StxError(f"conv: input has {x.shape[-1]} channels, weight expects {w.shape[1]}")
            ctx.cin_true, ctx.x_prepadded = w.shape[1], True
            w = pad_weight_channels(w, x.shape[-1])
        elif not transposed and x.shape[-1] % 8 != 0:
            # GEMM-K (input channels) in steps of 8: zero-pad odd widths (CFNet's 65- and 33-channel cascade volumes)
            ctx.cin_true = x.shape[-1]
            x, w = pad_input_channels(x, w)
        if _is_c1(w, ks, stride, transposed) and not want_stats:
            z, stats = conv3d_c1_forward(x, w.contiguous()), None
        elif transposed:
            Cout = w.shape[1]
            z, stats = deconv3d_forward(x, pack_weight(w, 2), Cout, want_stats=want_stats)
        else:
            Cout = w.shape[0]
            z, stats = conv3d_forward(x, pack_weight(w, 0), Cout, ks, stride, want_stats=want_stats)
        ctx.save_for_backward(x, w)
        ctx.cfg = (ks, stride, transposed)
        if stats is None:
            stats = x.new_empty(0)
        ctx.mark_non_differentiable(stats)
        return z, stats

    @staticmethod
    def backward(ctx, gz, _gstats):
        x, w = ctx.saved_tensors
        ks, stride, transposed = ctx.cfg
        pend = _take_pending_bn(gz)
        gx = gw = None
        B, D, H, W, Cin = x.shape
        if pend is not None:
            # gz is the gradient BEHIND this convolution's BatchNorm (BnActFn deferred its apply pass): the march weight
            # gradient forms dz itself and writes it for the data gradient below
            if (ctx.needs_input_grad[1] and not transposed and ks == 3 and stride == 1 and ctx.cin_true is None
                    and get_lib().raw("stx_conv3d_wgrad_bn_supported")(B, D, H, W, Cin, w.shape[0])):
                gw_pre, gz = conv3d_wgrad_bn(x, gz, pend)
                gw_pre = gw_pre.reshape(w.shape)
            else:
                gw_pre, gz = None, pend.materialize()
        else:
            gw_pre = None
        gz = gz.contiguous()
        if transposed:
            Ci, Co = w.shape[0], w.shape[1]
            if ctx.needs_input_grad[0]:   # stride-2 conv of gz with the deconv weight read as [Cout'][Cin']
                gx, _ = conv3d_forward(gz, pack_weight(w, 0), Ci, 3, 2)
            if ctx.needs_input_grad[1]:   # the wgrad kernel works on 32-channel pairs: pad CFNet's 16-wide deconvs
                gz_w = gz if Co % 32 == 0 else _pad_channels(gz, (Co + 31) // 32 * 32)
                x_w = x if Ci % 32 == 0 else _pad_channels(x, (Ci + 31) // 32 * 32)
                gw = conv3d_wgrad(gz_w, x_w, 3, 2)[:Ci, :Co].reshape(w.shape)
        else:
            Co, Ci = w.shape[0], w.shape[1]
            c1 = _is_c1(w, ks, stride, transposed)
            gz_k, w_k = gz, w
            if Co % 8 != 0 and not c1:    # pad GEMM-K to 8
                Cp = (Co + 7) // 8 * 8
                gz_k = _pad_channels(gz, Cp)
                w_k = w.new_zeros(Cp, *w.shape[1:])
                w_k[:Co] = w
            if ctx.needs_input_grad[0]:
                if c1:                    # classifier tail Conv3d(32 -> 1): streaming VALU kernel
                    gx = conv3d_c1_dgrad(gz, w.contiguous(), Ci)
                elif stride == 1 and Ci > 128:
                    # the conv kernels produce at most 128 output channels per launch: wide inputs (PCWNet's 192-channel
                    # fusion convs) get their gradient in slices of the weight's input-channel axis
                    parts = []
                    for lo in range(0, Ci, 128):
                        hi = min(Ci, lo + 128)
                        g, _ = conv3d_forward(gz_k, pack_weight(w_k[:, lo:hi].contiguous(), 1), hi - lo, ks, 1)
                        parts.append(g)
                    gx = torch.cat(parts, -1)
                elif stride == 1:
                    gx, _ = conv3d_forward(gz_k, pack_weight(w_k, 1), Ci, ks, 1)
                else:                     # stride-2 dgrad = transposed conv of gz (its kernel takes GEMM-K in steps of 32)
                    if Co % 32:
                        Cp = (Co + 31) // 32 * 32
                        gz_k = _pad_channels(gz, Cp)
                        w_k = w.new_zeros(Cp, *w.shape[1:])
                        w_k[:Co] = w
                    gx, _ = deconv3d_forward(gz_k, pack_weight(w_k, 2), Ci, out_dims=(D, H, W))
            if gw_pre is not None:
                gw = gw_pre
            elif ctx.needs_input_grad[1]:
                if c1:
                    gw = conv3d_c1_wgrad(x, gz)
                else:
                    gz_w = gz if Co % 32 == 0 else _pad_channels(gz, (Co + 31) // 32 * 32)
                    x_w = x if Ci % 32 == 0 else _pad_channels(x, (Ci + 31) // 32 * 32)
                    gw = conv3d_wgrad(x_w, gz_w, ks, stride)[:Co, :Ci].reshape(w.shape)
        if ctx.cin_true is not None:
            if gx is not None and not ctx.x_prepadded:
                gx = gx[..., :ctx.cin_true].contiguous()
            if gw is not None:
                gw = gw[:, :ctx.cin_true].contiguous()
        return gx, gw, None, None, None, None


class SharedInputConvsFn(torch.autograd.Function):
    """Raw outputs (+ BN partial sums) of SEVERAL plain convolutions of ONE activation -- the tensors with more than one
    convolution consumer in the reference graphs: the hourglass input (conv1 stride 2 + redir1 1x1x1, gwcnet.py:85,103) together
    with the classifier head that reads the same volume (gwcnet.py:192-195), conv2's output (conv3 + redir2).  Forward = the
    same launches ConvRawFn makes.  Backward: autograd would sum the consumers' input gradients with one volume-sized `add`
    per extra consumer (three passes over a 212 MB volume each: 0.65 ms per GwcNet_GC train step); here every gradient after
    the first is accumulated in the epilogue of the kernel that produces it (`residual` operand: one extra read).  Order: the
    3x3x3 stride-1 layer first (its march kernel keeps its straight-line epilogue), then stride 2, then 1x1x1.
    cfgs: one (ks, stride, want_stats) per convolution; weights [Cout, Cin, k, k, k], Cin % 8 == 0, Cout % 8 == 0."""

    @staticmethod
    def forward(ctx, x, cfgs, *ws):
        _chk(x, "x", 5)
        outs = []
        for (ks, stride, want), w in zip(cfgs, ws):
            if x.shape[-1] != w.shape[1] or w.shape[1] % 8 or w.shape[0] % 8 or _is_c1(w, ks, stride, False):
                raise StxError(f"SharedInputConvsFn: weight {tuple(w.shape)} on an input of {x.shape[-1]} channels")
            z, stats = conv3d_forward(x, pack_weight(w, 0), w.shape[0], ks, stride, want_stats=want)
            if stats is None:
                stats = x.new_empty(0)
            ctx.mark_non_differentiable(stats)
            outs += [z, stats]
        ctx.save_for_backward(x, *ws)
        ctx.cfgs = cfgs
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        x, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        B, D, H, W, Ci = x.shape
        gws = [None] * len(ws)
        gzs = []
        for k, g in enumerate(grads[0::2]):
            pend = _take_pending_bn(g)
            if pend is not None:                       # (see ConvRawFn.backward)
                ks, stride, _ = ctx.cfgs[k]
                if (ctx.needs_input_grad[2 + k] and ks == 3 and stride == 1
                        and get_lib().raw("stx_conv3d_wgrad_bn_supported")(B, D, H, W, Ci, ws[k].shape[0])):
                    gw, g = conv3d_wgrad_bn(x, g, pend)
                    gws[k] = gw.reshape(ws[k].shape)
                else:
                    g = pend.materialize()
            gzs.append(None if g is None else g.contiguous())
        gx = None
        if ctx.needs_input_grad[0]:
            rank = lambda k: (0 if ctx.cfgs[k][0] == 3 and ctx.cfgs[k][1] == 1 else (1 if ctx.cfgs[k][1] == 2 else 2))
            for k in sorted((k for k in range(len(ws)) if gzs[k] is not None), key=rank):
                (ks, stride, _), w, gz = ctx.cfgs[k], ws[k], gzs[k]
                if stride == 1:
                    gx, _ = conv3d_forward(gz, pack_weight(w, 1), Ci, ks, 1, residual=gx)
                else:                     # stride-2 dgrad = transposed conv of gz (its kernel takes GEMM-K in steps of 32)
                    Co = w.shape[0]
                    gz_k, w_k = gz, w
                    if Co % 32:
                        Cp = (Co + 31) // 32 * 32
                        gz_k = _pad_channels(gz, Cp)
                        w_k = w.new_zeros(Cp, *w.shape[1:])
                        w_k[:Co] = w
                    gx, _ = deconv3d_forward(gz_k, pack_weight(w_k, 2), Ci, out_dims=(D, H, W), residual=gx)
        for k, (w, gz) in enumerate(zip(ws, gzs)):
            if gz is None or not ctx.needs_input_grad[2 + k] or gws[k] is not None:
                continue
            ks, stride, _ = ctx.cfgs[k]
            Co = w.shape[0]
            gz_w = gz if Co % 32 == 0 else _pad_channels(gz, (Co + 31) // 32 * 32)
            x_w = x if Ci % 32 == 0 else _pad_channels(x, (Ci + 31) // 32 * 32)
            gws[k] = conv3d_wgrad(x_w, gz_w, ks, stride)[:Co, :Ci].reshape(w.shape)
        return (gx, None, *gws)


# --------------------------------------------------------------------------------------- ConvTranspose3d(k4, s2, p1)
_D4_SEL = ((3, 1, 4), (4, 2, 0))       # per axis and output parity p: the k4 tap at offsets -1, 0, +1 (4 = the zero slice)


def embed_deconv4_weight(w):
    """ConvTranspose3d(k=4, s=2, p=1) weight [Cin, Cout, 4, 4, 4] -> the weight [8 Cout, Cin, 3, 3, 3] of ONE stride-1
    3x3x3 convolution whose output channel (4 pd + 2 ph + pw) Cout + c is parity class (pd, ph, pw) of output channel c:
    per axis  out[2j + p] = x[j] w[1 + p] + x[j - 1 + 2p] w[3 - 3p], i.e. offsets (-1, 0, +1) take taps (3, 1, -) for
    p = 0 and (-, 2, 0) for p = 1 (reference IGEVStereo/igev_stereo.py:44-51 `conv*_up`).  Pure indexing: differentiable."""
    Ci, Co = w.shape[0], w.shape[1]
    wz = torch.cat((w, w.new_zeros(Ci, Co, 1, 4, 4)), 2)
    wz = torch.cat((wz, wz.new_zeros(Ci, Co, 5, 1, 4)), 3)
    wz = torch.cat((wz, wz.new_zeros(Ci, Co, 5, 5, 1)), 4)
    parts = []
    for pd in (0, 1):
        for ph in (0, 1):
            for pw in (0, 1):
                sel = wz[:, :, list(_D4_SEL[pd])][:, :, :, list(_D4_SEL[ph])][:, :, :, :, list(_D4_SEL[pw])]
                parts.append(sel.transpose(0, 1))
    return torch.cat(parts, 0).contiguous()


class DepthToSpaceFn(torch.autograd.Function):
    """[B, D, H, W, 8 C] (parity-class-major channels) -> [B, 2D, 2H, 2W, C]; backward = the inverse map."""

    @staticmethod
    def forward(ctx, y):
        y = y.contiguous()
        _chk(y, "y", 5)
        B, D, H, W, C8 = y.shape
        C = C8 // 8
        out = torch.empty(B, 2 * D, 2 * H, 2 * W, C, dtype=torch.float32, device=y.device)
        _call("stx_depth_to_space", _p(y), _p(out), B, D, H, W, C, 0)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, D2, H2, W2, C = g.shape
        gy = torch.empty(B, D2 // 2, H2 // 2, W2 // 2, 8 * C, dtype=torch.float32, device=g.device)
        _call("stx_depth_to_space", _p(g), _p(gy), B, D2 // 2, H2 // 2, W2 // 2, C, 1)
        return gy


def _d4_chunks(Co):
    """Parity classes per launch: the convolution kernels produce at most 128 output channels."""
    per = max(1, min(8, 128 // Co))
    return [(c0, min(8, c0 + per)) for c0 in range(0, 8, per)]


def deconv4_raw(x, w, want_stats):
    """Autograd path of ConvTranspose3d(k4, s2, p1): raw output z [B, 2D, 2H, 2W, Cout] (+ BatchNorm partial rows).
    One (or a few, for 8 Cout > 128) stride-1 3x3x3 MFMA convolutions on the embedded weight + the class interleave;
    data and weight gradients flow back through the same kernels (ConvRawFn) and the indexing above."""
    Co = w.shape[1]
    if Co % 4:
        raise StxError(f"ConvTranspose3d(k4): {Co} output channels (need a multiple of 4)")
    w3 = embed_deconv4_weight(w)
    ys = [ConvRawFn.apply(x, w3[c0 * Co:c1 * Co], 3, 1, False, False)[0] for c0, c1 in _d4_chunks(Co)]
    z = DepthToSpaceFn.apply(ys[0] if len(ys) == 1 else torch.cat(ys, -1))
    return z, (bn_stats(z.detach()) if want_stats else z.new_empty(0))


@fp32_region
def deconv4_forward(x, w, scale=None, bias=None, relu=0, owner=None):
    """Inference path of ConvTranspose3d(k4, s2, p1) with the folded BatchNorm / activation in the convolution epilogue
    (scale / bias repeated per parity class), then the class interleave.  Packed weights cached on `owner`."""
    _chk(x, "x", 5)
    Co = w.shape[1]
    if Co % 4:
        raise StxError(f"ConvTranspose3d(k4): {Co} output channels (need a multiple of 4)")
    cache = owner.__dict__.setdefault("_stx_packed", {}) if (owner is not None and _CACHE_ENABLED) else None
    key = (owner._version, owner.data_ptr()) if owner is not None else None
    hit = cache.get("d4") if cache is not None else None
    if hit is not None and hit[0] == key:
        packed = hit[1]
    else:
        w3 = embed_deconv4_weight(w)
        packed = [pack_weight(w3[c0 * Co:c1 * Co].contiguous(), 0) for c0, c1 in _d4_chunks(Co)]
        if cache is not None:
            cache["d4"] = (key, packed)
    ys = []
    for (c0, c1), wp in zip(_d4_chunks(Co), packed):
        n = c1 - c0
        ys.append(conv3d_forward(x, wp, n * Co, 3, 1, None if scale is None else scale.repeat(n),
                                 None if bias is None else bias.repeat(n), None, relu)[0])
    B, D, H, W, _ = x.shape
    y = ys[0] if len(ys) == 1 else torch.cat(ys, -1)
    out = torch.empty(B, 2 * D, 2 * H, 2 * W, Co, dtype=torch.float32, device=x.device)
    _call("stx_depth_to_space", _p(y), _p(out), B, D, H, W, Co, 0)
    return out


# --------------------------------------------------------------------------------------- FeatureAtt gate
class GateFn(torch.autograd.Function):
    """cv [B, D, H, W, C] * sigmoid(att [B, H, W, C]) (reference IGEVStereo/submodule.py:238-240), one pass; backward one
    pass over (g, cv) with the gate's gradient summed over D in registers."""

    @staticmethod
    def forward(ctx, cv, att):
        out = gate(cv.detach(), att.detach())
        ctx.save_for_backward(cv, att)
        return out

    @staticmethod
    def backward(ctx, g):
        cv, att = ctx.saved_tensors
        g = g.contiguous()
        B, D, H, W, C = cv.shape
        gcv = torch.empty_like(cv) if ctx.needs_input_grad[0] else None
        gatt = torch.empty_like(att) if ctx.needs_input_grad[1] else None
        if gcv is not None or gatt is not None:
            _call("stx_gate_bwd", _p(g), _p(cv), _p(att), _p(gcv), _p(gatt), B, D, H * W, C)
        return gcv, gatt


@fp32_region
def gate(cv, att):
    """cv [B, D, H, W, C] (dense NDHWC) * sigmoid(att [B, H, W, C]) broadcast over D; differentiable."""
    if torch.is_grad_enabled() and (cv.requires_grad or att.requires_grad):
        return GateFn.apply(cv.contiguous(), att.contiguous())
    cv, att = cv.contiguous(), att.contiguous()
    _chk(cv, "cv", 5)
    _chk(att, "att", 4)
    B, D, H, W, C = cv.shape
    if att.shape != (B, H, W, C) or C % 4:
        raise StxError(f"gate: volume {tuple(cv.shape)} vs gate {tuple(att.shape)} (need [B,H,W,C], C % 4 == 0)")
    out = torch.empty_like(cv)
    _call("stx_gate_fwd", _p(cv), _p(att), _p(out), B, D, H * W, C)
    return out


# --------------------------------------------------------------------------------------- channels-last <-> channel-major
class ChannelMajorFn(torch.autograd.Function):
    """NCHW-logical tensor stored channels-last (dense [B, H, W, C]) -> the same values stored NCHW-contiguous, through the
    tiled transpose kernel (stx_transpose); backward: the inverse transpose, so the gradient goes back dense in the
    producer's layout (torch's `contiguous()` takes its generic strided copy both ways: 0.10 ms per 320 x 144 x 240 map, and
    its backward hands the producer an NCHW gradient that every channels-last consumer re-lays again)."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        out = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device)
        _call("stx_transpose", _p(x.permute(0, 2, 3, 1)), _p(out), B, H * W, C)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, C, H, W = g.shape
        gx = torch.empty((B, H, W, C), dtype=g.dtype, device=g.device)
        _call("stx_transpose", _p(g), _p(gx), B, C, H * W)
        return gx.permute(0, 3, 1, 2)


@fp32_region
def channel_major(x):
    """`x.contiguous()` for a 4-D NCHW-logical tensor: a no-op for NCHW-contiguous input, the transpose kernel for dense
    channels-last fp32 input whose H*W and C are multiples of 4, torch's copy otherwise."""
    if x.is_contiguous():
        return x
    if (x.dim() == 4 and x.dtype == torch.float32 and on_device(x) and x.is_contiguous(memory_format=torch.channels_last)
            and x.shape[1] % 4 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0):
        return ChannelMajorFn.apply(x)
    return x.contiguous()


# --------------------------------------------------------------------------------------- channel concatenation
class CatChannelsFn(torch.autograd.Function):
    """Dense channels-last tensors [..., C_k] -> [..., sum C_k] in one coalesced pass (stx_concat_channels); backward: one
    pass that writes the dense per-part gradients (stx_split_channels).  Reference: `torch.cat(..., dim=1)` of the feature
    extractors (gwcnet.py:59, acv.py:48)."""

    @staticmethod
    def forward(ctx, *parts):
        ctx.cs = [int(p.shape[-1]) for p in parts]
        nvox = parts[0].numel() // ctx.cs[0]
        out = parts[0].new_empty(*parts[0].shape[:-1], sum(ctx.cs))
        ptrs = [_p(p) for p in parts] + [None] * (4 - len(parts))
        cs = ctx.cs + [0] * (4 - len(parts))
        _call("stx_concat_channels", *ptrs, *cs, _p(out), nvox)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        nvox = g.numel() // g.shape[-1]
        outs = [g.new_empty(*g.shape[:-1], c) if need else None for c, need in zip(ctx.cs, ctx.needs_input_grad)]
        if any(o is not None for o in outs):
            # (a part that needs no gradient still gets a destination: the kernel writes every part)
            dst = [o if o is not None else g.new_empty(*g.shape[:-1], c) for o, c in zip(outs, ctx.cs)]
            ptrs = [_p(o) for o in dst] + [None] * (4 - len(dst))
            cs = ctx.cs + [0] * (4 - len(dst))
            _call("stx_split_channels", _p(g), *ptrs, *cs, nvox)
        return tuple(outs)


@fp32_region
def cat_channels(parts):
    """Concatenate 2-4 dense channels-last tensors of equal leading shape along their LAST (channel) axis; every width a
    multiple of 4.  Differentiable."""
    parts = [p.contiguous() for p in parts]
    if not 2 <= len(parts) <= 4:
        raise StxError(f"cat_channels: {len(parts)} parts (2 to 4)")
    lead = parts[0].shape[:-1]
    for p in parts:
        if p.shape[:-1] != lead or p.shape[-1] % 4 or p.dtype != torch.float32 or p.device != parts[0].device:
            raise StxError(f"cat_channels: parts {[tuple(q.shape) for q in parts]} (equal leading shape, fp32, widths % 4 == 0)")
    return CatChannelsFn.apply(*parts)


# --------------------------------------------------------------------------------------- 2-D convolution (feature CNN)
def conv2d_supported(conv):
    """Whether an nn.Conv2d is one csrc/conv2d.hip serves in BOTH directions: 3x3, stride 1, padding 1, dilation 1, dense, no
    bias, 32 or 64 input AND output channels (the data gradient's GEMM-K is the layer's output width)."""
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None and conv.in_channels in (32, 64) and conv.out_channels in (32, 64))


def _ohwi(w):
    """[Co][Ci][3][3] parameter -> dense [Co][3][3][Ci] (a view when the parameter is stored channels_last, as the extractors
    keep theirs: features2d.channels_last_weights_)."""
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def conv2d_forward(x, w, dgrad=False, groups=1, want_stats=False):
    """x dense [B, H, W, Cin]; w the layer's parameter [Co][Ci][3][3].  dgrad=False: conv(x, w) -> [B, H, W, Co]; dgrad=True: x is
    the output gradient [B, H, W, Co] -> the input gradient [B, H, W, Ci].  want_stats: also the per-workgroup (sum, sum of
    squares) rows of the raw output, [rows, 2, C] ([groups, rows / groups, 2, C]: per-view statistics) -> bn_finalize."""
    _chk(x, "x", 4)
    wo = _ohwi(w)
    _chk(wo, "weight", 4)
    B, H, W, Cin = x.shape
    Co, Ci = w.shape[0], w.shape[1]
    Kin, Nout = (Co, Ci) if dgrad else (Ci, Co)
    if Cin != Kin or B % groups:
        raise StxError(f"conv2d: input {tuple(x.shape)} against weight {tuple(w.shape)} (dgrad={dgrad}, groups={groups})")
    out = torch.empty(B, H, W, Nout, dtype=torch.float32, device=x.device)
    part = None
    if want_stats:
        rows = int(get_lib().raw("stx_conv2d_stat_rows")(groups))
        part = torch.empty(rows, 2, Nout, dtype=torch.float32, device=x.device)
    _call("stx_conv2d_fwd", _p(x), _p(wo), _p(out), _p(part), B, H, W, Kin, Nout, int(dgrad), groups)
    if part is not None and groups > 1:
        part = part.view(groups, rows // groups, 2, Nout)
    return out, part


class Conv2dFn(torch.autograd.Function):
    """z = conv2d(x, w) (3x3, stride 1, padding 1) of the 2-D feature CNN's BasicBlocks on csrc/conv2d.hip: forward and data
    gradient hand-written, the weight gradient stays MIOpen's (aten.convolution_backward).  x NCHW-logical channels_last;
    returns z as dense [B, H, W, Co] and the BatchNorm statistics rows of z (not differentiable)."""

    @staticmethod
    def forward(ctx, x, w, groups):
        xl = x.permute(0, 2, 3, 1)
        if not xl.is_contiguous():
            xl = xl.contiguous()
        z, part = conv2d_forward(xl, w, False, groups, True)
        ctx.save_for_backward(x, w)
        ctx.mark_non_differentiable(part)
        return z, part

    @staticmethod
    def backward(ctx, gz, _gpart):
        x, w = ctx.saved_tensors
        gz = gz.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = conv2d_forward(gz, w, True)[0].permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = torch.ops.aten.convolution_backward(gz.permute(0, 3, 1, 2), x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                     (False, True, False))[1]
        return gx, gw, None


# --------------------------------------------------------------------------------------- batch norm
def bn_finalize(partials, count, gamma, beta, running_mean, running_var, momentum, eps, groups=1):
    """partials [rows, 2, C] ([groups, rows, 2, C] for groups > 1) -> scale, shift, mean, invstd ([C] each, [groups, C] for
    groups > 1: one launch, running statistics updated slab after slab)."""
    C = partials.shape[-1]
    out = torch.empty(4, groups, C, dtype=torch.float32, device=partials.device)
    _call("stx_bn_finalize_groups", _p(partials), partials.shape[-3], C, float(count), _p(gamma), _p(beta),
          _p(running_mean), _p(running_var), float(momentum), float(eps), _p(out), groups)
    return list(out.unbind(0)) if groups > 1 else list(out[:, 0].unbind(0))   # scale, shift, mean, invstd


@fp32_region
def bn_stats(z, groups=1):
    """Per-workgroup (sum, sum of squares) rows of a dense channels-last activation [..., C] -> [rows, 2, C]
    ([groups, rows, 2, C] for groups > 1: the leading axis of z splits into `groups` equal slabs with their own
    statistics): the batch statistics input of bn_finalize for producers without a fused epilogue (the 2-D feature
    CNN's MIOpen convolutions)."""
    _chk(z, "z")
    C = z.shape[-1]
    nvox = z.numel() // C // groups
    if groups < 1 or z.shape[0] % groups:
        raise StxError(f"bn_stats: leading axis {z.shape[0]} does not split into {groups} groups")
    rows = get_lib().raw("stx_bn_stats_rows")(nvox, C)
    if rows <= 0:
        raise StxError(f"bn_stats: C={C} unsupported")
    part = torch.empty(groups, rows, 2, C, dtype=torch.float32, device=z.device)
    _call("stx_bn_stats", _p(z), _p(part), nvox, C, groups)
    return part if groups > 1 else part[0]


def _sync_bn_partials(partials, count, group, world):
    """SyncBatchNorm forward: sum the conv-epilogue partials (sum z, sum z^2 per channel) over the replicas.
    The local rows are reduced in fp64, all-reduced in fp64 (2C values: latency-bound, RCCL), and handed to
    stx_bn_finalize as two fp32 rows hi + lo so that its fp64 accumulation sees the exact global sums
    (E[z^2] - mean^2 cancels badly on single-precision totals).  Equal per-replica batches are assumed, as
    DistributedSampler(drop_last=True) guarantees (trainer_torchrun.py:130-136)."""
    import torch.distributed as dist
    tot = partials.double().sum(0)                 # [2, C]
    dist.all_reduce(tot, group=group)
    hi = tot.float()
    lo = (tot - hi.double()).float()
    return torch.stack((hi, lo)).contiguous(), count * world


@fp32_region
def bn_apply(z1, scale1, shift1, z2=None, scale2=None, shift2=None, relu=False, groups=1):
    out = torch.empty_like(z1)
    C = z1.shape[-1]
    _call("stx_bn_apply", _p(z1), _p(scale1), _p(shift1), _p(z2), _p(scale2), _p(shift2), _p(out),
          z1.numel() // C // groups, C, int(relu), groups)
    return out


class BnActFn(torch.autograd.Function):
    """y = act(BN1(z1) [+ BN2(z2)] [+ residual]) with batch statistics (train) or running statistics
    (eval with grad).  `bn1`/`bn2` are dicts: gamma, beta, running_mean, running_var, momentum, eps,
    training, partials (conv-epilogue sums) and count.
    groups > 1: the leading axis of z1 (z2, residual) splits into `groups` equal slabs with their OWN batch statistics
    (partials [groups, rows, 2, C], count = voxels per slab); the running statistics are updated slab after slab, as
    separate forward calls would (the two views of the 2-D feature CNN, reference gwcnet.py:172-173)."""

    @staticmethod
    def forward(ctx, z1, gamma1, beta1, z2, gamma2, beta2, residual, relu, bn1, bn2, groups=1, defer=False):
        G = int(groups)
        # defer: z1 is the raw output of a 3x3x3 stride-1 convolution whose backward node folds this BatchNorm's
        # backward-apply pass into its weight-gradient kernel (conv_block sets it; see _PendingBn)
        ctx.defer = bool(defer)

        def affine(gamma, beta, bn):
            if bn["training"]:
                partials, count = bn["partials"], bn["count"]
                if bn.get("sync"):
                    if G > 1:
                        raise StxError("BnActFn: SyncBatchNorm with groups is not wired (SyncBN runs the views separately)")
                    partials, count = _sync_bn_partials(partials, count, *bn["sync"])
                return bn_finalize(partials, count, gamma, beta, bn["running_mean"], bn["running_var"], bn["momentum"],
                                   bn["eps"], G)                                              # [C] each ([G, C] for G > 1)
            invstd = torch.rsqrt(bn["running_var"] + bn["eps"])
            scale = gamma * invstd
            res = (scale, beta - bn["running_mean"] * scale, bn["running_mean"].clone(), invstd)
            return res if G == 1 else [t.unsqueeze(0).expand(G, -1).contiguous() for t in res]

        relu = int(relu)                                  # activation code: 0 none, 1 ReLU, 2 Mish
        if relu == 2 and residual is not None:            # (before anything is launched or any running statistic moves)
            raise StxError("BnActFn: Mish with a plain residual is not wired (no model of the family uses it)")
        if G < 1 or z1.shape[0] % G:
            raise StxError(f"BnActFn: leading axis {z1.shape[0]} does not split into {G} groups")
        sc1, sh1, m1, i1 = affine(gamma1, beta1, bn1)
        two = z2 is not None
        if two:
            sc2, sh2, m2, i2 = affine(gamma2, beta2, bn2)
            y = bn_apply(z1, sc1, sh1, z2, sc2, sh2, relu, G)
        else:
            sc2 = sh2 = m2 = i2 = None
            y = bn_apply(z1, sc1, sh1, residual, None, None, relu, G)
        ctx.relu, ctx.two, ctx.has_res, ctx.groups = relu, two, residual is not None, G
        ctx.train1 = bn1["training"]
        ctx.train2 = bn2["training"] if two else False
        ctx.sync = bn1.get("sync") or (bn2.get("sync") if two else None)
        # ReLU mask for the backward pass: a block without a plain residual recomputes sign(y) from z1 / z2 and the
        # scale / shift vectors used above (bit-identical expression, operands the backward kernels read anyway), so
        # the activated volume is not kept alive by this node and not re-read by its two backward passes
        ctx.remask = bool(relu) and residual is None
        keep_y = y if (relu and not ctx.remask) else None
        ctx.save_for_backward(z1, gamma1, m1, i1, z2, gamma2, m2, i2, keep_y, sc1 if ctx.remask else None,
                              sh1 if ctx.remask else None, sc2 if (ctx.remask and two) else None,
                              sh2 if (ctx.remask and two) else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        z1, gamma1, m1, i1, z2, gamma2, m2, i2, y, sc1, sh1, sc2, sh2 = ctx.saved_tensors
        gy = gy.contiguous()
        G = ctx.groups
        C = z1.shape[-1]
        nvox = z1.numel() // C // G                       # voxels per group
        lib = get_lib()
        NB = lib.raw("stx_bn_reduce_blocks")()
        part = _WS.get("bnred", G * NB * 3 * C, z1.device)
        # (groups > 1: one more slab = the total over the groups, written by the same launch)
        sums_all = torch.empty(G + (1 if G > 1 else 0), 3, C, dtype=torch.float32, device=z1.device)
        sums = sums_all[:G]
        _call("stx_bn_bwd_reduce2", _p(gy), _p(y), _p(z1), _p(m1), _p(i1), _p(z2) if ctx.two else None,
              _p(m2) if ctx.two else None, _p(i2) if ctx.two else None, _p(sc1), _p(sh1), _p(sc2), _p(sh2), _p(part),
              _p(sums_all), nvox, C, int(ctx.relu), G)
        if not ctx.needs_input_grad[0] and not ctx.two and not ctx.has_res:
            # nobody differentiates z1 (frozen convolution weight AND an input without gradient, e.g. a BatchNorm-only
            # fine-tune behind a frozen backbone): only the gamma / beta gradients are wanted -- no apply pass, and in
            # particular no deferred record that no convolution backward would ever consume
            tot = sums_all[G] if G > 1 else sums_all[0]
            return (None, tot[1], tot[0], None, None, None, None, None, None, None, None, None)
        if (ctx.defer and not ctx.two and not ctx.has_res and ctx.relu in (0, 1) and ctx.train1 and not ctx.sync and G == 1
                and (not ctx.relu or ctx.remask)):
            # the record travels on a FRESH alias of the gradient: only the node that receives this very return value (z1's
            # producer) can see it -- `gy` itself may also be delivered to other nodes by an add's fan-out
            alias = gy.view_as(gy)
            pend = _PendingBn()
            pend.g, pend.z, pend.scale, pend.shift, pend.mean, pend.invstd = alias, z1, sc1, sh1, m1, i1
            pend.gamma, pend.sums, pend.inv_n, pend.act = gamma1, sums, 1.0 / nvox, int(ctx.relu)
            if pend.scale is None:                     # (no activation: the mask operands were not saved; any finite pair does)
                pend.scale = pend.shift = m1
            _defer_pending_bn(alias, pend)
            tot = sums_all[0]
            return (alias, tot[1], tot[0], None, None, None, None, None, None, None, None, None)
        dz1 = torch.empty_like(z1)
        dz2 = torch.empty_like(z2) if ctx.two else None
        gres = torch.empty_like(z1) if (ctx.has_res and ctx.relu) else None
        use = sums
        if ctx.sync:
            # SyncBatchNorm: the centering terms are means over ALL replicas; the kernel divides by the local voxel
            # count, so hand it (sum over replicas) / world.  Parameter gradients stay local (DDP averages them).
            import torch.distributed as dist
            use = sums.clone()
            dist.all_reduce(use, group=ctx.sync[0])
            use /= ctx.sync[1]
        if not ctx.train1 or (ctx.two and not ctx.train2):
            # running-stat BN: no centering terms (mixed train/eval pairs are not used by these models)
            use = torch.zeros_like(sums)
        _call("stx_bn_bwd_apply2", _p(gy), _p(y), _p(z1), _p(m1), _p(i1), _p(gamma1), _p(z2) if ctx.two else None,
              _p(m2) if ctx.two else None, _p(i2) if ctx.two else None, _p(gamma2) if ctx.two else None, _p(sc1), _p(sh1),
              _p(sc2), _p(sh2), _p(use), _p(dz1), _p(dz2), _p(gres), nvox, C, int(ctx.relu), G)
        if ctx.has_res and not ctx.relu:
            gres = gy
        tot = sums_all[G] if G > 1 else sums_all[0]       # gamma / beta are shared between the groups
        # (beta2's gradient equals beta1's; it gets its OWN memory: two parameters whose .grad alias one buffer are scaled twice
        # by every in-place pass over the gradients -- GradScaler.unscale_, clip_grad_norm_ -- found by tests/test_amp.py)
        return (dz1, tot[1], tot[0], dz2, tot[2] if ctx.two else None, tot[0].clone() if ctx.two else None, gres,
                None, None, None, None, None)


# --------------------------------------------------------------------------------------- activations
class MishFn(torch.autograd.Function):
    """y = x * tanh(softplus(x)) (PCWNet/CFNet `Mish` / `FMish`, models/PCWNet/submodule.py:11-18,178-190)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _chk(x, "x")
        y = torch.empty_like(x)
        _call("stx_mish_fwd", _p(x), _p(y), x.numel())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        _call("stx_mish_bwd", _p(gy), _p(x), _p(gx), x.numel())
        return gx


@fp32_region
def mish(x):
    """Mish on a dense fp32 tensor with numel % 4 == 0 (every channels-last activation of these models)."""
    if torch.is_grad_enabled() and x.requires_grad:
        return MishFn.apply(x)
    x = x.contiguous()
    _chk(x, "x")
    y = torch.empty_like(x)
    _call("stx_mish_fwd", _p(x), _p(y), x.numel())
    return y


# --------------------------------------------------------------------------------------- cost volume
def _cv_shapes(Lg, Lc, num_groups):
    ref = Lg if Lg is not None else Lc
    B, _, H, W = ref.shape
    Cg = Lg.shape[1] if Lg is not None else 0
    Cc = Lc.shape[1] if Lc is not None else 0
    G = num_groups if Lg is not None else 0
    return B, H, W, Cg, G, Cc


@fp32_region
def cost_volume_forward(Lg, Rg, Lc, Rc, maxdisp, num_groups, mask_left=True, scale=None):
    for n, t in (("ref gwc", Lg), ("tgt gwc", Rg), ("ref concat", Lc), ("tgt concat", Rc)):
        _chk(t, n, 4)
    B, H, W, Cg, G, Cc = _cv_shapes(Lg, Lc, num_groups)
    if G:
        assert Cg % G == 0   # reference models/GwcNet/submodule.py:46
    vol = torch.empty(B, maxdisp, H, W, G + 2 * Cc, dtype=torch.float32, device=(Lg if Lg is not None else Lc).device)
    _call("stx_cost_volume_fwd", _p(Lg), _p(Rg), Cg, G, _p(Lc), _p(Rc), Cc, _p(scale), _p(vol), B, H, W, maxdisp,
          int(mask_left))
    return vol


class CostVolumeFn(torch.autograd.Function):
    """Fused gwc + concat volume, NDHWC output [B, D, H, W, G + 2*Cc]."""

    @staticmethod
    def forward(ctx, Lg, Rg, Lc, Rc, maxdisp, num_groups, mask_left):
        ctx.save_for_backward(Lg, Rg)
        ctx.cfg = (maxdisp, num_groups, mask_left, None if Lc is None else Lc.shape)
        return cost_volume_forward(Lg, Rg, Lc, Rc, maxdisp, num_groups, mask_left)

    @staticmethod
    def backward(ctx, gvol):
        Lg, Rg = ctx.saved_tensors
        maxdisp, num_groups, mask_left, cshape = ctx.cfg
        gvol = gvol.contiguous()
        B, D, H, W, CT = gvol.shape
        Cg = Lg.shape[1] if Lg is not None else 0
        G = num_groups if Lg is not None else 0
        Cc = cshape[1] if cshape is not None else 0
        gLg = torch.empty_like(Lg) if G else None
        gRg = torch.empty_like(Rg) if G else None
        gLc = gvol.new_empty(cshape) if Cc else None
        gRc = gvol.new_empty(cshape) if Cc else None
        _call("stx_cost_volume_bwd", _p(gvol), _p(Lg), _p(Rg), Cg, G, Cc, _p(gLg), _p(gRg), _p(gLc), _p(gRc), B, H, W,
              D, int(mask_left))
        return gLg, gRg, gLc, gRc, None, None, None


class GroupNormalizeFn(torch.autograd.Function):
    """y = out_scale * x / max(||x||_2 over each group's channels, 1e-12), per pixel (stx_group_normalize_fwd / _bwd):
    F.normalize(x.view(B, G, C/G, H, W), dim=2) of FoundationStereo's group-wise correlation (submodule.py:388-397)."""

    @staticmethod
    def forward(ctx, x, num_groups, out_scale):
        x = x.contiguous()
        _chk(x, "x", 4)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        _call("stx_group_normalize_fwd", _p(x), _p(y), B, C, num_groups, H * W, float(out_scale))
        ctx.save_for_backward(x)
        ctx.cfg = (num_groups, float(out_scale))
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        _call("stx_group_normalize_bwd", _p(x), _p(gy), _p(gx), B, C, ctx.cfg[0], H * W, ctx.cfg[1])
        return gx, None, None


@fp32_region
def group_normalize(x, num_groups, out_scale=1.0):
    """[B, C, H, W] -> every group of C / num_groups channels scaled to unit L2 norm at every pixel (times out_scale)."""
    assert x.dim() == 4 and x.shape[1] % num_groups == 0, f"C:{x.shape[1]}, num_groups:{num_groups}"   # submodule.py:390
    return GroupNormalizeFn.apply(channel_major(x), num_groups, out_scale)


@fp32_region
def cost_volume(Lg, Rg, Lc, Rc, maxdisp, num_groups, mask_left=True, normalize=False):
    """normalize=True: FoundationStereo's volume (submodule.py:388-413) -- cosine similarity per group (unit-norm groups,
    group SUM): the two gwc maps go through group_normalize first, the left one scaled by C/G so that the builder's group
    mean is the sum."""
    if normalize and Lg is not None:
        Lg = group_normalize(Lg, num_groups, float(Lg.shape[1] // num_groups))
        Rg = group_normalize(Rg, num_groups, 1.0)
    ts = [channel_major(t) if t is not None else None for t in (Lg, Rg, Lc, Rc)]
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts):
        return CostVolumeFn.apply(*ts, maxdisp, num_groups, mask_left)
    return cost_volume_forward(*ts, maxdisp, num_groups, mask_left)


class SampledVolumeFn(torch.autograd.Function):
    """CFNet cascade-stage volume (stx_sampled_volume_fwd / _bwd): NDHWC [B, S, H, W, CTp], channels = G group
    correlations | Cc left concat | Cc warped right concat | the hypothesis itself | zero pad to a multiple of 8."""

    @staticmethod
    def forward(ctx, Lg, Rg, Lc, Rc, samples, num_groups):
        B, H, W, Cg, G, Cc = _cv_shapes(Lg, Lc, num_groups)
        S = samples.shape[1]
        CTp = (G + 2 * Cc + 1 + 7) // 8 * 8
        vol = torch.empty(B, S, H, W, CTp, dtype=torch.float32, device=samples.device)
        _call("stx_sampled_volume_fwd", _p(Lg), _p(Rg), Cg, G, _p(Lc), _p(Rc), Cc, _p(samples), _p(vol), B, H, W, S, CTp)
        ctx.save_for_backward(Lg, Rg, samples)
        ctx.cfg = (G, Cg, Cc, None if Lc is None else Lc.shape)
        return vol

    @staticmethod
    def backward(ctx, gvol):
        Lg, Rg, samples = ctx.saved_tensors
        G, Cg, Cc, cshape = ctx.cfg
        gvol = gvol.contiguous()
        B, S, H, W, CTp = gvol.shape
        gLg = torch.empty_like(Lg) if G else None
        gRg = torch.empty_like(Rg) if G else None
        gLc = gvol.new_empty(cshape) if Cc else None
        gRc = gvol.new_empty(cshape) if Cc else None
        _call("stx_sampled_volume_bwd", _p(gvol), _p(Lg), _p(Rg), Cg, G, Cc, _p(samples), _p(gLg), _p(gRg), _p(gLc),
              _p(gRc), B, H, W, S, CTp)
        return gLg, gRg, gLc, gRc, None, None


@fp32_region
def sampled_volume(Lg, Rg, Lc, Rc, samples, num_groups):
    """Cascade-stage cost volume from per-pixel disparity hypotheses `samples` [B, S, H, W] (integer-valued floats, no
    gradient): see include/stx_hip.h.  The channel axis is padded with zeros to a multiple of 8; `conv_block` accepts
    such a volume for a convolution whose weight has the un-padded input width."""
    ts = [channel_major(t) if t is not None else None for t in (Lg, Rg, Lc, Rc)]
    for n, t in (("ref gwc", ts[0]), ("tgt gwc", ts[1]), ("ref concat", ts[2]), ("tgt concat", ts[3])):
        _chk(t, n, 4)
    samples = samples.detach().contiguous()
    _chk(samples, "samples", 4)
    if ts[0] is not None:
        assert ts[0].shape[1] % num_groups == 0   # reference models/CFNet/submodule.py:165
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts):
        return SampledVolumeFn.apply(*ts, samples, num_groups)
    return SampledVolumeFn.forward(_NoCtx(), *ts, samples, num_groups)


class _NoCtx:
    """Stand-in for the autograd context when a Function's forward is used without a graph."""

    def save_for_backward(self, *a):
        pass


# --------------------------------------------------------------------------------------- PCWNet / CFNet 2-D helpers (row f-1)
class WarpFn(torch.autograd.Function):
    """reference PCWNet/submodule.py:137-176 (`warp`): stx_warp_fwd / stx_warp_bwd."""

    @staticmethod
    def forward(ctx, x, disp):
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        _call("stx_warp_fwd", _p(x), _p(disp), _p(out), B, C, H, W)
        ctx.save_for_backward(x, disp)
        return out

    @staticmethod
    def backward(ctx, g):
        x, disp = ctx.saved_tensors
        B, C, H, W = x.shape
        g = g.contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gd = torch.empty_like(disp) if ctx.needs_input_grad[1] else None
        _call("stx_warp_bwd", _p(g), _p(x), _p(disp), _p(gx), _p(gd), B, C, H, W)
        return gx, gd


@fp32_region
def warp(x, disp):
    """x [B,C,H,W] sampled at column w - disp[B,1,H,W] (bilinear, the reference's grid), zeroed where the footprint leaves
    the image.  Differentiable in x and disp like the reference."""
    x, disp = x.contiguous(), disp.contiguous()
    _chk(x, "x", 4)
    _chk(disp, "disp", 4)
    if disp.shape != (x.shape[0], 1, x.shape[2], x.shape[3]):
        raise StxError(f"warp: disp {tuple(disp.shape)} does not match x {tuple(x.shape)}")
    if torch.is_grad_enabled() and (x.requires_grad or disp.requires_grad):
        return WarpFn.apply(x, disp)
    return WarpFn.forward(_NoCtx(), x, disp)


class CorrVolumeFn(torch.autograd.Function):
    """reference PCWNet/submodule.py:121-135 (`build_corrleation_volume`): stx_corr_volume_fwd / _bwd."""

    @staticmethod
    def forward(ctx, ref, tgt, maxdisp, groups):
        B, C, H, W = ref.shape
        vol = torch.empty(B, groups, 2 * maxdisp + 1, H, W, dtype=torch.float32, device=ref.device)
        _call("stx_corr_volume_fwd", _p(ref), _p(tgt), _p(vol), B, C, H, W, maxdisp, groups)
        ctx.save_for_backward(ref, tgt)
        ctx.cfg = (maxdisp, groups)
        return vol

    @staticmethod
    def backward(ctx, g):
        ref, tgt = ctx.saved_tensors
        B, C, H, W = ref.shape
        g = g.contiguous()
        gr = torch.empty_like(ref) if ctx.needs_input_grad[0] else None
        gt = torch.empty_like(tgt) if ctx.needs_input_grad[1] else None
        _call("stx_corr_volume_bwd", _p(g), _p(ref), _p(tgt), _p(gr), _p(gt), B, C, H, W, *ctx.cfg)
        return gr, gt, None, None


@fp32_region
def corr_volume(ref, tgt, maxdisp, groups):
    """[B,C,H,W] x 2 -> [B, groups, 2*maxdisp+1, H, W] (see include/stx_hip.h for the slice semantics)."""
    ref, tgt = ref.contiguous(), tgt.contiguous()
    _chk(ref, "ref", 4)
    _chk(tgt, "tgt", 4)
    if ref.shape != tgt.shape:
        raise StxError(f"corr_volume: ref {tuple(ref.shape)} vs tgt {tuple(tgt.shape)}")
    if torch.is_grad_enabled() and (ref.requires_grad or tgt.requires_grad):
        return CorrVolumeFn.apply(ref, tgt, int(maxdisp), int(groups))
    return CorrVolumeFn.forward(_NoCtx(), ref, tgt, int(maxdisp), int(groups))


class DisparityVarianceFn(torch.autograd.Function):
    """reference CFNet/submodule.py:128-140 (`disparity_variance`, `disparity_variance_confidence`)."""

    @staticmethod
    def forward(ctx, x, disp, samples):
        B, D = x.shape[0], x.shape[1]
        HW = x.shape[2] * x.shape[3]
        out = torch.empty(B, 1, x.shape[2], x.shape[3], dtype=torch.float32, device=x.device)
        _call("stx_disparity_variance_fwd", _p(x), _p(disp), _p(samples), _p(out), B, D, HW)
        ctx.save_for_backward(x, disp, samples)
        return out

    @staticmethod
    def backward(ctx, g):
        x, disp, samples = ctx.saved_tensors
        B, D = x.shape[0], x.shape[1]
        HW = x.shape[2] * x.shape[3]
        g = g.contiguous()
        need = ctx.needs_input_grad
        gx = torch.empty_like(x) if need[0] else None
        gd = torch.empty_like(disp) if need[1] else None
        gs = torch.empty_like(samples) if (samples is not None and need[2]) else None
        if gx is None and gd is None and gs is None:
            return None, None, None
        _call("stx_disparity_variance_bwd", _p(g), _p(x), _p(disp), _p(samples), _p(gx), _p(gd), _p(gs), B, D, HW)
        return gx, gd, gs


@fp32_region
def disparity_variance(x, disp, samples=None):
    """sum_d x_d (d - disp)^2 (samples None) or sum_d x_d (disp - samples_d)^2 -> [B,1,H,W]."""
    x, disp = x.contiguous(), disp.contiguous()
    _chk(x, "x", 4)
    _chk(disp, "disparity", 4)
    if disp.shape != (x.shape[0], 1, x.shape[2], x.shape[3]):
        raise StxError(f"disparity_variance: disparity {tuple(disp.shape)} does not match x {tuple(x.shape)}")
    if samples is not None:
        samples = samples.contiguous()
        _chk(samples, "samples", 4)
        if samples.shape != x.shape:
            raise StxError(f"disparity_variance: samples {tuple(samples.shape)} vs x {tuple(x.shape)}")
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, disp, samples)):
        return DisparityVarianceFn.apply(x, disp, samples)
    return DisparityVarianceFn.forward(_NoCtx(), x, disp, samples)


class AcVolumeFn(torch.autograd.Function):
    """ACVNet attention concat volume: softmax(att, dim=2) * concat_volume (acv.py:196), left half
    unmasked (ACVNet/submodule.py:180-191).  prob: [B, D, H, W] softmax probabilities."""

    @staticmethod
    def forward(ctx, Lc, Rc, prob, maxdisp):
        ctx.save_for_backward(Lc, Rc, prob)
        ctx.maxdisp = maxdisp
        return cost_volume_forward(None, None, Lc, Rc, maxdisp, 0, mask_left=False, scale=prob)

    @staticmethod
    def backward(ctx, gvol):
        Lc, Rc, prob = ctx.saved_tensors
        gvol = gvol.contiguous()
        B, D, H, W, CT = gvol.shape
        Cc = Lc.shape[1]
        gprob = torch.empty_like(prob)
        gL, gR = torch.empty_like(Lc), torch.empty_like(Rc)
        if D * W * 4 <= 150 * 1024:
            # one pass over the gradient volume (the [D][W] image of gprob lives in LDS)
            _call("stx_ac_volume_bwd", _p(gvol), _p(Lc), _p(Rc), _p(prob), _p(gL), _p(gR), _p(gprob), B, Cc, H, W, D, 0)
            return gL, gR, gprob, None
        _call("stx_cost_volume_scale_bwd", _p(gvol), _p(Lc), _p(Rc), _p(gprob), B, Cc, H, W, D, 0)
        scaled = torch.empty_like(gvol)
        _call("stx_scale_channels", _p(gvol), _p(prob), _p(scaled), B * D * H * W, CT)
        _call("stx_cost_volume_bwd", _p(scaled), None, None, 0, 0, Cc, None, None, _p(gL), _p(gR), B, H, W, D, 0)
        return gL, gR, gprob, None


@fp32_region
def ac_volume(Lc, Rc, prob, maxdisp):
    Lc, Rc, prob = Lc.contiguous(), Rc.contiguous(), prob.contiguous()
    if torch.is_grad_enabled() and (Lc.requires_grad or Rc.requires_grad or prob.requires_grad):
        return AcVolumeFn.apply(Lc, Rc, prob, maxdisp)
    return cost_volume_forward(None, None, Lc, Rc, maxdisp, 0, mask_left=False, scale=prob)


class DwConvHWFn(torch.autograd.Function):
    """Depth-wise (1,3,3) convolution with per-channel-quad dilation on an NDHWC volume.
    w: [C, 9]; dil: int32 [C/4] device tensor."""

    @staticmethod
    def forward(ctx, x, w, dil):
        _chk(x, "x", 5)
        B, D, H, W, C = x.shape
        out = torch.empty_like(x)
        _call("stx_dwconv_hw_fwd", _p(x), _p(w), _p(dil), _p(out), B, D, H, W, C, 0)
        ctx.save_for_backward(x, w, dil)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w, dil = ctx.saved_tensors
        gy = gy.contiguous()
        B, D, H, W, C = x.shape
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _call("stx_dwconv_hw_fwd", _p(gy), _p(w), _p(dil), _p(gx), B, D, H, W, C, 1)
        if ctx.needs_input_grad[1]:
            ws = _WS.get("dwwgrad", get_lib().raw("stx_dwconv_hw_wgrad_workspace_floats")(C), x.device)
            gw = torch.empty_like(w)
            _call("stx_dwconv_hw_wgrad", _p(x), _p(gy), _p(dil), _p(gw), _p(ws), B, D, H, W, C)
        return gx, gw, None


@fp32_region
def dwconv_hw(x, w, dil):
    w = w.contiguous()
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return DwConvHWFn.apply(x, w, dil)
    B, D, H, W, C = x.shape
    out = torch.empty_like(x)
    _call("stx_dwconv_hw_fwd", _p(x), _p(w), _p(dil), _p(out), B, D, H, W, C, 0)
    return out


# --------------------------------------------------------------------------------------- head
class HeadFn(torch.autograd.Function):
    """trilinear upsample -> softmax over D -> soft-argmin, fused. cost: [B, D', H', W'] dense.
    align_corners=False: GwcNet / PSMNet / ACVNet heads; True: PCWNet / CFNet heads."""

    @staticmethod
    def forward(ctx, cost, maxdisp, H, W, align_corners=False):
        _chk(cost, "cost", 4)
        B, Dc, Hc, Wc = cost.shape
        disp = torch.empty(B, H, W, dtype=torch.float32, device=cost.device)
        stats = torch.empty(B, H, W, 2, dtype=torch.float32, device=cost.device)
        if align_corners:
            _call("stx_head_fwd2", _p(cost), _p(disp), _p(stats), B, Dc, Hc, Wc, maxdisp, H, W, 1)
        else:
            _call("stx_head_fwd", _p(cost), _p(disp), _p(stats), B, Dc, Hc, Wc, maxdisp, H, W)
        ctx.save_for_backward(cost, disp, stats)
        ctx.cfg = (maxdisp, H, W, bool(align_corners))
        return disp

    @staticmethod
    def backward(ctx, g):
        cost, disp, stats = ctx.saved_tensors
        maxdisp, H, W, ac = ctx.cfg
        B, Dc, Hc, Wc = cost.shape
        gc = torch.empty_like(cost)
        g = g.contiguous()      # keep the dense copy alive across the launch
        ws = _WS.get("headbwd", get_lib().raw("stx_head_bwd_workspace_floats")(B, Dc, H, W), cost.device)
        if ac:
            _call("stx_head_bwd2", _p(g), _p(cost), _p(disp), _p(stats), _p(gc), _p(ws), B, Dc, Hc, Wc, maxdisp, H, W, 1)
        else:
            _call("stx_head_bwd", _p(g), _p(cost), _p(disp), _p(stats), _p(gc), _p(ws), B, Dc, Hc, Wc, maxdisp, H, W)
        return gc, None, None, None, None


@fp32_region
def regression_head(cost, maxdisp, H, W, align_corners=False):
    """cost [B, D', H', W'] (or [B,1,D',H',W'] / NDHWC with C=1) -> disparity [B, H, W]."""
    if cost.dim() == 5:
        cost = cost.reshape(cost.shape[0], *_squeeze_c(cost))
    return HeadFn.apply(cost.contiguous(), maxdisp, H, W, align_corners)


def _squeeze_c(cost):
    """Shape of a 5-D single-channel cost ([B,1,D,H,W] or [B,D,H,W,1]) without the channel axis."""
    if cost.shape[1] == 1:
        return tuple(cost.shape[2:])
    if cost.shape[-1] == 1:
        return tuple(cost.shape[1:4])
    raise StxError(f"regression_head: expected a single-channel cost, got {tuple(cost.shape)}")


@fp32_region
def softargmax(x, maxdisp, keepdim):
    """sum_d d * x[b,d,h,w] (disparity_regression / disparityregression / softargmax estimator)."""
    assert len(x.shape) == 4   # reference models/GwcNet/submodule.py:24
    _chk(x.contiguous(), "x", 4)
    x = x.contiguous()
    B, D, H, W = x.shape
    if D != maxdisp:
        raise StxError(f"disparity_regression: volume has {D} disparities, maxdisp={maxdisp}")
    if torch.is_grad_enabled() and x.requires_grad:
        # linear map: gradient is d broadcast -- keep autograd simple with a tiny custom function
        out = _SoftArgmaxFn.apply(x)
    else:
        out = torch.empty(B, H, W, dtype=torch.float32, device=x.device)
        _call("stx_softargmax_fwd", _p(x), _p(out), B, D, H * W)
    return out.unsqueeze(1) if keepdim else out


class _SoftArgmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, D, H, W = x.shape
        out = torch.empty(B, H, W, dtype=torch.float32, device=x.device)
        _call("stx_softargmax_fwd", _p(x), _p(out), B, D, H * W)
        ctx.D = D
        return out

    @staticmethod
    def backward(ctx, g):
        d = torch.arange(ctx.D, dtype=g.dtype, device=g.device).view(1, ctx.D, 1, 1)
        return g.unsqueeze(1) * d


@fp32_region
def argmax_disparity(x):
    _chk(x.contiguous(), "x", 4)
    x = x.contiguous()
    B, D, H, W = x.shape
    out = torch.empty(B, 1, H, W, dtype=torch.int64, device=x.device)
    _call("stx_argmax_fwd", _p(x), _p(out), B, D, H * W)
    return out


class _ModalFn(torch.autograd.Function):
    """Modal estimators with autograd: the support mask is a constant of the graph (the reference multiplies by
    `mask.data` / boolean masks), so d out / d x_k = m_k (k - out) / S; the forward pass saves the support and its
    mass per pixel ([B,5,H,W]) for the backward kernel."""

    @staticmethod
    def forward(ctx, x, kind):
        B, D, H, W = x.shape
        out = torch.empty(B, 1, H, W, dtype=torch.float32, device=x.device)
        aux = torch.empty(B, 5, H, W, dtype=torch.float32, device=x.device)
        _call("stx_modal_fwd", _p(x), _p(out), _p(aux), B, D, H * W, kind)
        ctx.save_for_backward(out, aux)
        ctx.D = D
        return out

    @staticmethod
    def backward(ctx, g):
        out, aux = ctx.saved_tensors
        B, _, H, W = out.shape
        g = g.contiguous()
        gx = torch.empty(B, ctx.D, H, W, dtype=torch.float32, device=out.device)
        _call("stx_modal_bwd", _p(g), _p(out), _p(aux), _p(gx), B, ctx.D, H * W)
        return gx, None


def _modal_estimator(kind, x, maxdisp):
    assert len(x.shape) == 4
    name = ("unimodal", "dominant_modal")[kind]
    x = x.contiguous()
    _chk(x, "x", 4)
    B, D, H, W = x.shape
    if D != maxdisp:
        raise StxError(f"{name}_disparity_estimator: volume has {D} disparities, maxdisp={maxdisp}")
    if torch.is_grad_enabled() and x.requires_grad:
        return _ModalFn.apply(x, kind)
    out = torch.empty(B, 1, H, W, dtype=torch.float32, device=x.device)
    _call("stx_modal_fwd", _p(x), _p(out), None, B, D, H * W, kind)
    return out


@fp32_region
def unimodal_disparity(x, maxdisp):
    """Expectation over the mode containing the arg-max (unimodal_disparity_estimator.py:4-25) -> [B,1,H,W];
    differentiable w.r.t. x inside the (constant) mode mask, like the reference."""
    return _modal_estimator(0, x, maxdisp)


@fp32_region
def dominant_modal_disparity(x, maxdisp):
    """Expectation over the heavier of the two main modes of the blurred volume
    (dominant_modal_disparity_estimator.py:35-54) -> [B,1,H,W]; differentiable like the reference."""
    return _modal_estimator(1, x, maxdisp)


class _SplitModeFn(torch.autograd.Function):
    """mode = x * mask with the mask a constant of the graph (boolean in the reference): d mode / d x = mask."""

    @staticmethod
    def forward(ctx, x):
        B, D, H, W = x.shape
        mode = torch.empty_like(x)
        mask = torch.empty(B, D, H, W, dtype=torch.bool, device=x.device)
        _call("stx_split_mode", _p(x), _p(mode), _p(mask), B, D, H * W)
        ctx.save_for_backward(mask)
        ctx.mark_non_differentiable(mask)
        return mode, mask

    @staticmethod
    def backward(ctx, g, _gmask):
        (mask,) = ctx.saved_tensors
        return g * mask


@fp32_region
def split_mode(x, maxdisp=192):
    """(mode, mask) of loss_functions/split_mode.py:9-35: the support of the mode around the per-pixel arg-max of the
    probability volume x [B, D, H, W] (twin of the modal estimators' mask, on the raw volume) as a bool tensor, and
    mode = x * mask.  One kernel, the volume read once (the reference builds ~12 full-volume temporaries)."""
    assert len(x.shape) == 4
    x = x.contiguous()
    _chk(x, "x", 4)
    N, D, H, W = x.shape
    assert D == maxdisp
    if torch.is_grad_enabled() and x.requires_grad:
        return _SplitModeFn.apply(x)
    mode = torch.empty_like(x)
    mask = torch.empty(N, D, H, W, dtype=torch.bool, device=x.device)
    _call("stx_split_mode", _p(x), _p(mode), _p(mask), N, D, H * W)
    return mode, mask


@fp32_region
def softmax_over_d(x):
    """x [B, D, H, W] -> softmax over D."""
    _chk(x, "x", 4)
    B, D, H, W = x.shape
    y = torch.empty_like(x)
    _call("stx_softmax_d_fwd", _p(x), _p(y), B, D, H * W)
    return y
