"""SURVEY.md 8f rank 4, the "small hourglass": the IGEV-family initial-disparity aggregation (igev_stereo.py:23-100,
148-151, 206-213; IGEVStereo/submodule.py:9-38, 228-241) on the HIP engine.

* the oracle's restatement vs the reference's OWN classes (tests/golden/igev_agg.npz, made by make_golden_igev_agg.py);
* new kernel entry points vs stock torch ops (depth_to_space, FeatureAtt gate, LeakyReLU in the fused BN passes / the conv
  epilogue), emulator (CPU) and gfx950 (`-m gpu`);
* the drop-in modules (`BasicConv`, `FeatureAtt`, `hourglass`, `IGEVCostAggregation`) vs the oracle: eval, and a full
  train step with every parameter / input gradient and the BatchNorm running statistics; state-dict keys vs the reference's.
"""
import contextlib
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_oracle as O
from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor
from tests.backends import be, ndhwc, ptr  # noqa: F401
from tests.golden.igev_agg_config import B, FEAT_CH, H4, MAXDISP, W4, fill

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gold():
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(G, "igev_agg.npz")).items()}


def _inputs(requires_grad=False):
    ml, mr = synthetic_tensor((B, 96, H4, W4), 71), synthetic_tensor((B, 96, H4, W4), 72)
    feats = [synthetic_tensor((B, c, H4 >> i, W4 >> i), 73 + i) for i, c in enumerate(FEAT_CH)]
    if requires_grad:
        for t in [ml, mr] + feats:
            t.requires_grad_()
    return ml, mr, feats


def _filled_sd():
    from stereo_toolbox_amd.models.IGEVStereo import IGEVCostAggregation
    m = IGEVCostAggregation(MAXDISP)
    sd = fill(m.state_dict())
    m.load_state_dict(sd)
    return m, {k: v.clone() for k, v in sd.items()}


def _loss(geo, disp):
    gt = synthetic_tensor(tuple(disp.shape), 80, lo=0.0, hi=float(MAXDISP // 4 - 1)).to(disp.device)
    return F.smooth_l1_loss(disp, gt) + 0.1 * (geo * synthetic_tensor(tuple(geo.shape), 81).to(geo.device)).mean()


def _sliced(g):
    return g[:4] if g.numel() > 8192 else g


def close(a, b, tol, what=""):
    err = (a - b).abs().max().item()
    assert err <= tol * (1.0 + b.abs().max().item()), f"{what}: {err:.3e} vs scale {b.abs().max().item():.3e}"


# ------------------------------------------------------------------------------------------------ oracle pinned to the reference
def test_state_dict_keys_match_reference():
    m, _ = _filled_sd()
    with open(os.path.join(G, "state_dict_keys_igev_agg.json")) as f:
        ref = json.load(f)["IGEVCostAggregation"]
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref


def test_oracle_matches_reference_fixture():
    g = _gold()
    _, sd = _filled_sd()
    with torch.no_grad():
        ml, mr, feats = _inputs()
        geo, disp = O.igev_cost_aggregation(sd, ml, mr, feats, MAXDISP)
    close(geo[:, :, ::2, ::2, ::2], g["eval_geo"], 1e-5, "eval geo")
    close(disp, g["eval_disp"], 1e-5, "eval disp")
    rsd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    ml, mr, feats = _inputs(True)
    (geo, disp), cx = O.igev_cost_aggregation(rsd, ml, mr, feats, MAXDISP, training=True, return_ctx=True)
    loss = _loss(geo, disp)
    loss.backward()
    # (train mode: the volume's 12-element means are summed in another order than the reference's loop -- 1e-7 -- and the
    #  batch-statistics BatchNorms and the x 40 classifier amplify that to a few 1e-4 px; a wrong formula would be O(1))
    close(geo.detach()[:, :, ::2, ::2, ::2], g["train_geo"], 1e-5, "train geo")
    close(disp.detach(), g["train_disp"], 1e-4, "train disp")
    assert abs(loss.item() - float(g["train_loss"])) < 1e-4
    for k, v in g.items():
        if k.startswith("grad_") and k[5:] in rsd:
            close(_sliced(rsd[k[5:]].grad), v, 2e-4, k)
        if k.startswith("stat_") and k[5:] in cx.new_stats:
            close(cx.new_stats[k[5:]], v, 1e-6, k)
    close(ml.grad[:, ::8, ::2, ::2], g["grad_match_left"], 2e-4, "grad match_left")
    close(mr.grad[:, ::8, ::2, ::2], g["grad_match_right"], 2e-4, "grad match_right")
    close(feats[0].grad[:, ::8, ::2, ::2], g["grad_feat0"], 2e-4, "grad feat0")
    close(feats[3].grad[:, ::16], g["grad_feat3"], 2e-4, "grad feat3")
    # unit cases: the k4 transposed convolution block and the gate, forward + backward
    usd = {"conv.weight": torch.empty(16, 8, 4, 4, 4), "bn.weight": torch.empty(8), "bn.bias": torch.empty(8),
           "bn.running_mean": torch.empty(8), "bn.running_var": torch.empty(8), "bn.num_batches_tracked": torch.zeros((), dtype=torch.long)}
    fill_state_dict(usd, seed=99)
    usd = {("up." + k): (v.requires_grad_() if v.is_floating_point() and "running" not in k else v) for k, v in usd.items()}
    x = synthetic_tensor((1, 16, 3, 5, 9), 90).requires_grad_()
    y = O.igev_basic_conv(O.Ctx(usd, True), x, "up", deconv=True)
    y.backward(synthetic_tensor(tuple(y.shape), 91))
    close(y.detach(), g["up_y"], 1e-5, "up y")
    close(x.grad, g["up_gx"], 1e-5, "up gx")
    close(usd["up.conv.weight"].grad[:4], g["up_gw"], 1e-5, "up gw")
    close(usd["up.bn.weight"].grad, g["up_gbn"], 1e-5, "up gbn")


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("shape", [(2, 3, 4, 5, 8), (1, 2, 1, 7, 16), (1, 4, 3, 2, 4)])
def test_depth_to_space(be, shape):
    """stx_depth_to_space both ways vs the index definition (exact: a permutation)."""
    Bn, D, H, W, C = shape
    y = synthetic_tensor((Bn, D, H, W, 8 * C), 11)
    want = y.view(Bn, D, H, W, 2, 2, 2, C).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(Bn, 2 * D, 2 * H, 2 * W, C)
    out = be.empty(Bn, 2 * D, 2 * H, 2 * W, C)
    be.call("stx_depth_to_space", ptr(be.dev(y)), ptr(out), Bn, D, H, W, C, 0)
    assert torch.equal(out.cpu(), want)
    back = be.empty(Bn, D, H, W, 8 * C)
    be.call("stx_depth_to_space", ptr(out), ptr(back), Bn, D, H, W, C, 1)
    assert torch.equal(back.cpu(), y)


def test_embedded_deconv4_weight_is_the_transposed_convolution():
    """ops.embed_deconv4_weight + depth-to-space == F.conv_transpose3d(k4, s2, p1) (CPU, stock ops: the identity the
    kernels rely on), including the weight gradient through the indexing."""
    from stereo_toolbox_amd.ops import embed_deconv4_weight
    x = synthetic_tensor((2, 8, 3, 4, 5), 21)
    w = synthetic_tensor((8, 4, 4, 4, 4), 22).requires_grad_()
    want = F.conv_transpose3d(x, w, None, stride=2, padding=1)
    y8 = F.conv3d(x, embed_deconv4_weight(w), None, 1, 1)                     # [B, 8*Co, D, H, W]
    Bn, _, D, H, W = y8.shape
    got = y8.view(Bn, 2, 2, 2, 4, D, H, W).permute(0, 4, 5, 1, 6, 2, 7, 3).reshape(Bn, 4, 2 * D, 2 * H, 2 * W)
    assert (got - want).abs().max().item() < 1e-5
    gy = synthetic_tensor(tuple(want.shape), 23)
    (gw_ref,) = torch.autograd.grad(want, w, gy, retain_graph=True)
    (gw,) = torch.autograd.grad(got, w, gy)
    assert (gw - gw_ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("shape", [(2, 5, 4, 6, 8), (1, 12, 3, 7, 48)])
def test_gate_fwd_bwd(be, shape):
    Bn, D, H, W, C = shape
    cv = synthetic_tensor((Bn, D, H, W, C), 31).requires_grad_()
    att = (synthetic_tensor((Bn, H, W, C), 32) * 3).requires_grad_()
    want = cv * torch.sigmoid(att).unsqueeze(1)
    g = synthetic_tensor(tuple(want.shape), 33)
    want.backward(g)
    out = be.empty(Bn, D, H, W, C)
    be.call("stx_gate_fwd", ptr(be.dev(cv.detach())), ptr(be.dev(att.detach())), ptr(out), Bn, D, H * W, C)
    close(out.cpu(), want.detach(), 1e-6, "gate fwd")
    gcv, gatt = be.empty(Bn, D, H, W, C), be.empty(Bn, H, W, C)
    be.call("stx_gate_bwd", ptr(be.dev(g)), ptr(be.dev(cv.detach())), ptr(be.dev(att.detach())), ptr(gcv), ptr(gatt), Bn, D,
            H * W, C)
    close(gcv.cpu(), cv.grad, 1e-6, "gate gcv")
    close(gatt.cpu(), att.grad, 1e-5, "gate gatt")


def test_leaky_relu_activation_code(be):
    """Activation code 3 = LeakyReLU(0.01): stx_bn_apply and the two y-free backward passes vs stock torch ops on
    z * scale + shift; and the conv epilogue (inference form) vs F.conv3d + affine + leaky_relu."""
    nvox, C = 600, 16
    z = synthetic_tensor((nvox, C), 41)
    sc, sh = synthetic_tensor((C,), 42, lo=0.5, hi=1.5), synthetic_tensor((C,), 43)
    y = be.empty(nvox, C)
    be.call("stx_bn_apply", ptr(be.dev(z)), ptr(be.dev(sc)), ptr(be.dev(sh)), None, None, None, ptr(y), nvox, C, 3, 1)
    want = F.leaky_relu(z * sc + sh, 0.01)
    close(y.cpu(), want, 1e-6, "bn_apply leaky")
    # backward of y = leaky(BN_train(z)): compare dz with autograd through batch_norm + leaky_relu
    zr = z.clone().requires_grad_()
    gam, bet = synthetic_tensor((C,), 44, lo=0.5, hi=1.5).requires_grad_(), synthetic_tensor((C,), 45).requires_grad_()
    yr = F.leaky_relu(F.batch_norm(zr, None, None, gam, bet, True, 0.1, 1e-5), 0.01)
    gy = synthetic_tensor((nvox, C), 46)
    yr.backward(gy)
    mean, var = z.mean(0), z.var(0, unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    scale, shift = gam.detach() * invstd, bet.detach() - mean * gam.detach() * invstd
    NB = be.raw("stx_bn_reduce_blocks")()
    part, sums = be.empty(NB, 3, C), be.empty(3, C)
    dz = be.empty(nvox, C)
    d = be.dev
    be.call("stx_bn_bwd_reduce2", ptr(d(gy)), None, ptr(d(z)), ptr(d(mean)), ptr(d(invstd)), None, None, None, ptr(d(scale)),
            ptr(d(shift)), None, None, ptr(part), ptr(sums), nvox, C, 3, 1)
    be.call("stx_bn_bwd_apply2", ptr(d(gy)), None, ptr(d(z)), ptr(d(mean)), ptr(d(invstd)), ptr(d(gam.detach())), None, None,
            None, None, ptr(d(scale)), ptr(d(shift)), None, None, ptr(sums), ptr(dz), None, None, nvox, C, 3, 1)
    close(dz.cpu(), zr.grad, 2e-5, "leaky BN dz")
    close(sums.cpu()[0], bet.grad, 2e-5, "leaky BN dbeta")
    close(sums.cpu()[1], gam.grad, 2e-5, "leaky BN dgamma")
    # conv epilogue, inference form
    x = synthetic_tensor((1, 8, 4, 5, 9), 47)
    w = synthetic_tensor((16, 8, 3, 3, 3), 48) * 0.2
    ref = F.leaky_relu(F.conv3d(x, w, None, 1, 1) * sc.view(1, C, 1, 1, 1) + sh.view(1, C, 1, 1, 1), 0.01)
    wp = be.empty(be.raw("stx_conv3d_packed_floats")(8, 16, 27))
    be.call("stx_conv3d_pack_weight", ptr(d(w)), ptr(wp), 16, 8, 27, 0)
    out = be.empty(1, 4, 5, 9, 16)
    be.call("stx_conv3d_fwd", ptr(d(ndhwc(x))), ptr(wp), ptr(out), ptr(d(sc)), ptr(d(sh)), None, None, 1, 4, 5, 9, 8, 16, 3, 1, 3)
    close(out.cpu().permute(0, 4, 1, 2, 3), ref, 1e-5, "conv epilogue leaky")


# ------------------------------------------------------------------------------------------------ drop-in modules
class Env:
    def __init__(self, name):
        self.name = name
        if name == "hip" and not torch.cuda.is_available():
            pytest.skip("no ROCm device")
        self.device = torch.device("cuda:0" if name == "hip" else "cpu")

    def ctx(self):
        if self.name == "emu":
            from tests.emu_util import emu_product_path
            return emu_product_path()
        return contextlib.nullcontext()


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def env(request):
    return Env(request.param)


def test_basic_conv_deconv4_and_feature_att_modules(env):
    """BasicConv(deconv k4 s2 p1 + BN + LeakyReLU, train mode) and FeatureAtt (eval) as nn.Modules with the reference's
    call convention (NCDHW in / out), forward + backward vs the fixture produced by the reference's own classes."""
    from stereo_toolbox_amd.models.IGEVStereo import BasicConv, FeatureAtt
    g = _gold()
    up = BasicConv(16, 8, deconv=True, is_3d=True, bn=True, relu=True, kernel_size=(4, 4, 4), padding=(1, 1, 1), stride=(2, 2, 2))
    usd = up.state_dict()
    fill_state_dict(usd, seed=99)
    up.load_state_dict(usd)
    up = up.to(env.device).train()
    x = synthetic_tensor((1, 16, 3, 5, 9), 90).to(env.device).requires_grad_()
    with env.ctx():
        y = up(x)
        y.backward(synthetic_tensor(tuple(y.shape), 91).to(env.device))
    assert y.shape == (1, 8, 6, 10, 18)
    close(y.detach().cpu(), g["up_y"], 1e-4, "up y")
    close(x.grad.cpu(), g["up_gx"], 1e-4, "up gx")
    close(up.conv.weight.grad.cpu()[:4], g["up_gw"], 1e-4, "up gw")
    close(up.bn.weight.grad.cpu(), g["up_gbn"], 1e-4, "up gbn")
    fa = FeatureAtt(8, 24)
    fsd = fa.state_dict()
    fill_state_dict(fsd, seed=98)
    fa.load_state_dict(fsd)
    fa = fa.to(env.device).eval()
    cv = synthetic_tensor((2, 8, 5, 4, 6), 92).to(env.device).requires_grad_()
    ft = synthetic_tensor((2, 24, 4, 6), 93).to(env.device).requires_grad_()
    with env.ctx():
        z = fa(cv, ft)
        z.backward(synthetic_tensor(tuple(z.shape), 94).to(env.device))
    close(z.detach().cpu(), g["fa_y"], 1e-5, "fa y")
    close(cv.grad.cpu(), g["fa_gcv"], 1e-5, "fa gcv")
    close(ft.grad.cpu(), g["fa_gfeat"], 1e-4, "fa gfeat")


def test_igev_cost_aggregation_eval_parity(env, parity_log):
    m, sd = _filled_sd()
    m = m.to(env.device).eval()
    ml, mr, feats = _inputs()
    with env.ctx(), torch.no_grad():
        geo, disp = m(ml.to(env.device), mr.to(env.device), [f.to(env.device) for f in feats])
    with torch.no_grad():
        rgeo, rdisp = O.igev_cost_aggregation(sd, ml, mr, feats, MAXDISP)
    assert geo.shape == rgeo.shape == (B, 8, MAXDISP // 4, H4, W4) and disp.shape == rdisp.shape == (B, 1, H4, W4)
    assert rdisp.std() > 0.2, "degenerate test output"
    e_disp = (disp.cpu() - rdisp).abs().max().item()
    parity_log(f"igev_agg_eval[{env.name}]", max_abs_init_disp=e_disp,
               max_abs_geo=(geo.cpu() - rgeo).abs().max().item(), geo_scale=rgeo.abs().max().item())
    assert e_disp < 1e-3                                            # the disparity bar of BASELINE.json (1/4-resolution pixels)
    close(geo.cpu(), rgeo, 1e-4, "geo_encoding_volume")


def test_igev_cost_aggregation_train_parity(env, parity_log):
    """One train step through every kernel of the aggregation (stride-2 / 1x1x1 / k4-transposed / 8..48-channel convolutions,
    LeakyReLU BN passes, gates, volume, classifier, softmax + regression): outputs, loss, EVERY parameter gradient, input
    gradients and BatchNorm running statistics vs the oracle; tolerances calibrated with the oracle's fp64 evaluation like
    the other train-step tests (batch-statistics BN amplifies fp32 rounding differences)."""
    m, sd = _filled_sd()
    m = m.to(env.device).train()
    ml, mr, feats = _inputs(True)
    dml, dmr = ml.detach().to(env.device).requires_grad_(), mr.detach().to(env.device).requires_grad_()
    dfe = [f.detach().to(env.device).requires_grad_() for f in feats]
    with env.ctx():
        geo, disp = m(dml, dmr, dfe)
        loss = _loss(geo, disp)
        loss.backward()
    rsd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    (rgeo, rdisp), cx = O.igev_cost_aggregation(rsd, ml, mr, feats, MAXDISP, training=True, return_ctx=True)
    rl = _loss(rgeo, rdisp)
    rl.backward()
    sd64 = {k: (v.double().requires_grad_("running" not in k) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    in64 = [t.detach().double().requires_grad_() for t in [ml, mr] + feats]
    g64, d64 = O.igev_cost_aggregation(sd64, in64[0], in64[1], in64[2:], MAXDISP, training=True)
    gt = synthetic_tensor(tuple(d64.shape), 80, lo=0.0, hi=float(MAXDISP // 4 - 1)).double()
    (F.smooth_l1_loss(d64, gt) + 0.1 * (g64 * synthetic_tensor(tuple(g64.shape), 81).double()).mean()).backward()
    e_prod = (disp.detach().cpu().double() - d64.detach()).abs().max().item()
    e_orc = (rdisp.detach().double() - d64.detach()).abs().max().item()
    assert e_prod < max(1e-3, 2.0 * e_orc), (e_prod, e_orc)
    assert abs(loss.item() - rl.item()) < 1e-4 * max(1.0, abs(rl.item()))
    worst, n = 0.0, 0
    for k, p in m.named_parameters():
        r, r64 = rsd[k].grad, sd64[k].grad
        assert p.grad is not None and r is not None, k
        scale = r.abs().max().item()
        e_p = (p.grad.cpu().double() - r64).abs().max().item()
        e_o = (r.double() - r64).abs().max().item()
        tol = max((1e-2 if env.name == "hip" else 2e-3) * scale, 3.0 * e_o) + 1e-7
        assert e_p <= tol, f"{k}: grad err {e_p:.3e} vs scale {scale:.3e} (oracle fp32 err {e_o:.3e})"
        worst = max(worst, e_p / (scale + 1e-12))
        n += 1
    for got, ref, r64, name in zip([dml, dmr] + dfe, [ml, mr] + feats, in64, ["match_left", "match_right", "f0", "f1", "f2", "f3"]):
        scale = ref.grad.abs().max().item()
        e_p = (got.grad.cpu().double() - r64.grad).abs().max().item()
        e_o = (ref.grad.double() - r64.grad).abs().max().item()
        assert e_p <= max(2e-3 * scale, 3.0 * e_o) + 1e-7, (name, e_p, e_o, scale)
    msd = m.state_dict()
    for k, v in cx.new_stats.items():
        assert (msd[k].cpu() - v).abs().max().item() < 1e-4 * max(1.0, v.abs().max().item()), k
    assert int(msd["cost_agg.conv2.1.bn.num_batches_tracked"]) == 1
    parity_log(f"igev_agg_train[{env.name}]", init_disp_vs_fp64=e_prod, oracle_fp32_vs_fp64=e_orc, worst_grad_rel_to_max=worst,
               tensors=n)
