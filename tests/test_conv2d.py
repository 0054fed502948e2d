"""The 2-D feature CNN's 3x3 stride-1 convolutions on the hand-written kernel (csrc/conv2d.hip) against stock fp32 torch
(`F.conv2d` on the CPU = what the reference's `convbn(in, out, 3, 1, 1, 1)` computes, models/GwcNet/gwcnet.py:12-42): forward,
data gradient (the same kernel on the flipped, channel-transposed weight), the fused BatchNorm statistics with per-view rows.
Runs on the host emulator build (CPU) and on the gfx950 library (-m gpu), through the C-ABI."""
import pytest
import torch
import torch.nn.functional as F

from tests.backends import be, ptr  # noqa: F401

# B, Cin, Cout, H, W, groups
CASES = [
    (1, 32, 32, 8, 16, 1),        # one tile
    (2, 64, 64, 19, 37, 2),       # ragged rows and columns, two views
    (2, 32, 48, 9, 50, 1),        # three output slices, runs crossing slices
    (4, 64, 16, 24, 33, 2),       # one slice, two images per view
    (1, 64, 128, 16, 16, 1),      # eight slices on two tiles: most workgroups idle
]


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _run(be, x, w, dgrad, groups, want_stats):
    """x: NCHW-logical input of the launch; w: the layer's parameter [Cout][Cin][3][3]."""
    Co, Ci = w.shape[:2]
    Kin, Nout = (Co, Ci) if dgrad else (Ci, Co)
    assert be.raw("stx_conv2d_supported")(Kin, Nout)
    wo = be.dev(w.permute(0, 2, 3, 1).contiguous())            # = the channels_last storage of the parameter
    B, _, H, W = x.shape
    xd = be.dev(_nhwc(x))
    out = be.empty(B, H, W, Nout)
    rows = int(be.raw("stx_conv2d_stat_rows")(groups))
    stats = be.empty(rows, 2, Nout) if want_stats else None
    be.call("stx_conv2d_fwd", ptr(xd), ptr(wo), ptr(out), ptr(stats) if want_stats else None, B, H, W, Kin, Nout, int(dgrad), groups)
    return out.cpu(), (stats.cpu() if want_stats else None), rows


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv2d_forward_and_statistics(be, case):
    B, Ci, Co, H, W, G = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.1
    ref = F.conv2d(x, w, None, 1, 1)
    out, stats, rows = _run(be, x, w, False, G, True)
    got = out.permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() + 1e-6, err
    # per-view sums of the raw output: rows [g * rows / G, (g + 1) * rows / G) belong to view g
    assert rows % G == 0
    st = stats.view(G, rows // G, 2, Co).double().sum(1)
    rv = ref.view(G, B // G, Co, H, W).double()
    s1, s2 = rv.sum((1, 3, 4)), (rv * rv).sum((1, 3, 4))
    assert (st[:, 0] - s1).abs().max().item() <= 1e-4 * s2.sqrt().max().item() + 1e-4
    assert (st[:, 1] - s2).abs().max().item() <= 1e-5 * s2.abs().max().item() + 1e-4


@pytest.mark.parametrize("case", [(1, 32, 32, 8, 16), (2, 64, 64, 19, 37), (1, 32, 64, 10, 20), (2, 64, 32, 9, 18)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_conv2d_data_gradient(be, case):
    B, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Ci, H, W, generator=g, requires_grad=True)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.1
    gy = torch.randn(B, Co, H, W, generator=g)
    F.conv2d(x, w, None, 1, 1).backward(gy)
    out, _, _ = _run(be, gy, w, True, 1, False)
    got = out.permute(0, 3, 1, 2)
    err = (got - x.grad).abs().max().item()
    assert err <= 2e-5 * x.grad.abs().max().item() + 1e-6, err


def test_conv2d_refuses_unsupported_shapes(be):
    from stereo_toolbox_amd._capi import StxError
    assert not be.raw("stx_conv2d_supported")(128, 128)
    assert not be.raw("stx_conv2d_supported")(64, 24)
    x = be.empty(1, 8, 16, 128, fill=0.0)
    wp = be.empty(9 * 128 * 128, fill=0.0)                    # (a weight buffer large enough for either refused call)
    out = be.empty(1, 8, 16, 128)
    with pytest.raises(StxError):
        be.call("stx_conv2d_fwd", ptr(x), ptr(wp), ptr(out), None, 1, 8, 16, 128, 128, 0, 1)
    x = be.empty(3, 8, 16, 32, fill=0.0)
    with pytest.raises(StxError):                      # batch 3 does not split into two views
        be.call("stx_conv2d_fwd", ptr(x), ptr(wp), ptr(out), None, 3, 8, 16, 32, 32, 0, 2)


# ------------------------------------------------------------------------------------ the product's wiring (features2d)
class _Env:
    def __init__(self, name):
        self.name = name
        if name == "hip" and not torch.cuda.is_available():
            pytest.skip("no ROCm device")
        self.device = torch.device("cuda:0" if name == "hip" else "cpu")

    def ctx(self):
        import contextlib
        if self.name == "emu":
            from tests.emu_util import emu_product_path
            return emu_product_path()
        return contextlib.nullcontext()


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def env(request):
    return _Env(request.param)


def test_basic_block_train_step_on_the_own_kernel_matches_the_stock_convolution(env, monkeypatch):
    """A train-mode BasicBlock (models/features2d.py = reference gwcnet.py:24-42) with its two 3x3 convolutions on csrc/conv2d.hip
    (forward, data gradient, fused BatchNorm statistics; the weight gradient stays aten's) against the same block with
    STX_FEAT2D_CONV=0 (stock F.conv2d + the stx_bn_stats pass): outputs, input / weight / BatchNorm gradients and running
    statistics, two views with per-view statistics."""
    from stereo_toolbox_amd import ops
    from stereo_toolbox_amd.models import features2d as F2

    def run(flag):
        monkeypatch.setenv("STX_FEAT2D_CONV", flag)
        torch.manual_seed(3)
        blk = F2.BasicBlock(32, 32, 1, None, 1, 1).to(env.device)
        F2.channels_last_weights_(blk)
        blk.train()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(4, 32, 12, 20, generator=g).to(env.device).contiguous(memory_format=torch.channels_last).requires_grad_()
        gy = torch.randn(4, 32, 12, 20, generator=g).to(env.device)
        calls = []
        orig = ops.Conv2dFn.apply
        monkeypatch.setattr(ops.Conv2dFn, "apply", staticmethod(lambda *a: (calls.append(1), orig(*a))[1]))
        with env.ctx(), F2.view_groups(2):
            y = blk(x)
            y.backward(gy)
        monkeypatch.setattr(ops.Conv2dFn, "apply", orig)
        out = {"y": y.detach().cpu(), "gx": x.grad.cpu()}
        out.update({"g:" + k: p.grad.cpu() for k, p in blk.named_parameters()})
        out.update({"b:" + k: v.detach().cpu().clone() for k, v in blk.named_buffers() if v.dtype.is_floating_point})
        return out, len(calls)

    own, n_own = run("1")
    ref, n_ref = run("0")
    assert n_own == 2 and n_ref == 0
    assert own.keys() == ref.keys()
    for k in ref:
        err = (own[k] - ref[k]).abs().max().item()
        assert err <= 2e-4 * ref[k].abs().max().item() + 1e-6, (k, err, ref[k].abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 64, 64, 144, 240), (2, 32, 32, 288, 480), (2, 64, 64, 96, 312), (4, 64, 64, 144, 240)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_conv2d_at_the_extractor_shapes(case):
    """The layer shapes of the BASELINE configs (SceneFlow 576x960 and KITTI 384x1248 pairs at 1/4 and 1/2 resolution, both views
    batched; cfg4's two pairs per GPU): forward, per-view statistics and data gradient against stock torch on the device
    (MIOpen, a different summation order: 2e-5 of the output's magnitude)."""
    from stereo_toolbox_amd import ops
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    B, Ci, Co, H, W = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, Ci, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, Co, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    ref = F.conv2d(x, w, None, 1, 1)
    rgx = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False))[0]
    z, part = ops.conv2d_forward(x.permute(0, 2, 3, 1), w, False, 2, True)
    gx, _ = ops.conv2d_forward(gy.permute(0, 2, 3, 1), w, True)
    assert (z.permute(0, 3, 1, 2) - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert (gx.permute(0, 3, 1, 2) - rgx).abs().max().item() <= 2e-5 * rgx.abs().max().item()
    st = part.double().sum(1).cpu()                                   # [2 views, 2, Co]
    rv = ref.double().view(2, B // 2, Co, H, W)
    s1, s2 = rv.sum((1, 3, 4)).cpu(), (rv * rv).sum((1, 3, 4)).cpu()
    assert (st[:, 0] - s1).abs().max().item() <= 1e-4 * s2.sqrt().max().item() + 1e-3
    assert (st[:, 1] - s2).abs().max().item() <= 1e-5 * s2.abs().max().item()
    # bit-reproducible launch to launch (fixed summation order, no atomics)
    z2, part2 = ops.conv2d_forward(x.permute(0, 2, 3, 1), w, False, 2, True)
    assert torch.equal(z, z2) and torch.equal(part, part2)
