"""The reference Trainer's mixed-precision path (VERDICT r5 "Missing" 1): trainer/trainer_torchrun.py:219 builds a
`torch.amp.GradScaler('cuda')` when `config.amp` is set, :274 runs the model and the loss under `torch.amp.autocast`, and
:286-294 do `scaler.scale(loss).backward()` / `unscale_` / `step` / `update`.  Under autocast the stock 2-D CNN hands fp16
(bf16 on the CPU emulator run) feature maps to the hand-written path, which computes in fp32 (`ops.fp32_region`): the
call used to die in `ops._chk` ("expected float32").

Checked, on the emulator build (`-m "not gpu"`, bf16 CPU autocast) and on the chip (`-m gpu`, fp16):
  * the train iteration runs, every prediction is fp32 and the loss finite; the optimizer step is taken (no inf / nan found);
  * WIRING, exact: the predictions under autocast are bit for bit the fp32 path's on the SAME low-precision features cast up
    -- AMP changes the features' rounding and nothing else on the hot path;
  * GradScaler scales THROUGH the fp32 kernels: two backward passes over ONE autocast forward, one scaled by 1024 -- the 3-D
    gradients divided by 1024 are bit for bit the plain ones (power-of-two scaling is exact in fp32; no atomics on this path);
  * against the full-fp32 step the predictions move by no more than the feature rounding explains (bound stated below);
  * outside autocast a low-precision tensor is still refused -- there is no fp16 kernel to fall back to.
"""
import pytest
import torch

from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor
from tests.test_models import Env, LOSS_W, _filled  # noqa: F401


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def env(request):
    return Env(request.param)


def _train_iteration(model, data, optimizer, scaler, amp, device, max_disp, low):
    """trainer_torchrun.py:264-303 with its one designed override (GwcNet returns a list of four predictions)."""
    from stereo_toolbox_amd.losses import masked_smooth_l1_multi
    optimizer.zero_grad()
    left, right, gt = (data[k].to(device) for k in ("left", "right", "gt_disp"))
    with torch.amp.autocast(device.type, enabled=amp, dtype=low):                   # :274
        preds = model(left, right)
        loss = masked_smooth_l1_multi(preds, gt.squeeze(1), max_disp, LOSS_W)
    if scaler is None:                                                                 # :282-285
        loss.backward()
        optimizer.step()
    else:                                                                              # :286-294
        scaler.scale(loss).backward()
        scaler.unscale_(optimizer)
        scaler.step(optimizer)
        scaler.update()
    return loss, preds


def _assert_gradients_do_not_alias(model):
    """No two parameters' .grad may overlap in memory: GradScaler.unscale_ and clip_grad_norm_ (trainer_torchrun.py:287-292)
    rescale every gradient IN PLACE, so a shared buffer is rescaled twice.  (Round 6: the two-BatchNorm block handed one row
    of its sums buffer to both betas -- `dresN.conv5.1.bias` / `redir2.1.bias`.)"""
    spans = sorted((p.grad.untyped_storage().data_ptr() + p.grad.storage_offset() * 4, p.grad.numel() * 4, k)
                   for k, p in model.named_parameters() if p.grad is not None)
    for (a, n, ka), (b, _, kb) in zip(spans, spans[1:]):
        assert a + n <= b, f"gradients of {ka} and {kb} share memory"


@pytest.mark.parametrize("ctor", ["GwcNet_GC", "ACVNet", "PSMNet"])
def test_train_iteration_under_autocast(env, ctor, parity_log):
    from stereo_toolbox_amd import models
    from stereo_toolbox_amd.models.features2d import run_pair
    dev = env.device
    low = torch.float16 if dev.type == "cuda" else torch.bfloat16
    if env.name == "emu":
        H, W, D, B = 16, 64, 32, 1
        if ctor != "GwcNet_GC":
            pytest.skip("emulator: GwcNet_GC only (CPU suite time); ACVNet on the GPU, PSMNet's 3-D path in test_psmnet_aggregate_under_autocast")
    else:
        if ctor == "PSMNet":
            pytest.skip("stock PyTorch-ROCm segfaults in F.batch_norm on the SPP branches' fp16 1x2 .. 8x16 maps under autocast "
                        "(GPU call B of round 6; MIOpen, not this package) -- PSMNet.aggregate under autocast is covered by "
                        "test_psmnet_aggregate_under_autocast")
        H, W, D, B = 64, 128, 64, 2
    data = {"left": synthetic_tensor((B, 3, H, W), 1), "right": synthetic_tensor((B, 3, H, W), 2),
            "gt_disp": synthetic_tensor((B, 1, H, W), 3, lo=0.0, hi=float(D - 2))}
    m, _ = _filled(getattr(models, ctor), D)
    m = m.to(dev).train()
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    scaler = torch.amp.GradScaler(dev.type, init_scale=1024.0)                     # :219 (power of two: scaling is exact in fp32)
    from stereo_toolbox_amd.losses import masked_smooth_l1_multi
    left, right, gt = (data[k].to(dev) for k in ("left", "right", "gt_disp"))
    with env.ctx():
        # trainer_torchrun.py:264-294 on ONE autocast forward (the stock fp16 2-D CNN is not run-to-run reproducible on the chip,
        # so the scaled and the plain backward below walk the same graph)
        opt.zero_grad()
        with torch.amp.autocast(dev.type, dtype=low):                                  # :274
            preds = m(left, right)
            loss = masked_smooth_l1_multi(preds, gt.squeeze(1), D, LOSS_W)
        assert all(p.dtype == torch.float32 for p in preds) and torch.isfinite(loss).item()
        scaler.scale(loss).backward(retain_graph=True)                                 # :286
        _assert_gradients_do_not_alias(m)
        g_scaled = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        opt.zero_grad()
        loss.backward()                                                                # the plain backward of the same graph
        hot = [k for k in g_scaled if not k.startswith(("feature_extraction.", "concatconv."))]
        assert len(hot) > 90
        named = dict(m.named_parameters())
        for k in hot:                       # GradScaler went THROUGH the fp32 kernels: power-of-two scaling is exact in fp32
            assert torch.equal(g_scaled[k] / 1024.0, named[k].grad), k
        for k, g in g_scaled.items():       # the scaled gradients back in place: the rest of the reference's sequence
            named[k].grad = g.clone()
        scaler.unscale_(opt)                                                           # :287
        assert all(torch.isfinite(p.grad).all().item() for p in m.parameters() if p.grad is not None)
        for k in hot:
            assert torch.equal(named[k].grad, g_scaled[k] / 1024.0), k                 # un-scaled once (no aliased gradient buffers)
        scaler.step(opt)                                                               # :290
        scaler.update()                                                                # :291
        assert scaler.get_scale() == 1024.0                                            # no inf / nan: the step was taken
        # wiring: autocast changes the features' rounding and nothing else on the hot path
        if hasattr(m, "aggregate") and ctor != "ACVNet":
            with torch.no_grad(), torch.amp.autocast(dev.type, dtype=low):
                fl, fr = run_pair(m.feature_extraction, data["left"].to(dev), data["right"].to(dev), True)
                some = next(iter(fl.values())) if isinstance(fl, dict) else fl
                assert some.dtype == low
                under = m.aggregate(fl, fr, H, W)
            up = (lambda f: {k: v.float() for k, v in f.items()}) if isinstance(fl, dict) else (lambda f: f.float())
            with torch.no_grad():
                plain = m.aggregate(up(fl), up(fr), H, W)
            for a, b in zip(under, plain):
                assert a.dtype == torch.float32 and torch.equal(a, b)
        if env.name == "emu":
            return
        # against the full-fp32 step: what the low-precision FEATURES move (fp16: 2^-11 relative)
        _, preds32 = _train_iteration(m, data, opt, None, False, dev, D, low)
    # a sanity bound, not a parity claim: the 2-D CNN itself ran in 8 (bf16) / 11 (fp16) significant bits, and these random-weight
    # networks amplify that (the exact statements are the three above)
    worst = max((a - b).abs().max().item() for a, b in zip(preds, preds32))
    mean = max((a - b).abs().mean().item() for a, b in zip(preds, preds32))
    parity_log(f"amp_vs_fp32_step[{env.name}-{ctor}]", low=str(low), worst_px=worst, mean_px=mean)
    assert mean < (D / 10 if low is torch.bfloat16 else D / 40), (worst, mean)


def test_psmnet_aggregate_under_autocast(env):
    """PSMNet.aggregate (concat volume -> PSM hourglasses -> cumulative heads) handed low-precision 32-channel features under
    autocast: fp32 predictions, bit for bit those of the fp32 call on the cast-up features; gradients reach the features in
    their own dtype."""
    from stereo_toolbox_amd import models
    dev = env.device
    low = torch.float16 if dev.type == "cuda" else torch.bfloat16
    D, h4, w4 = (32, 8, 16) if env.name == "emu" else (64, 16, 32)
    m, _ = _filled(models.PSMNet, D)
    m = m.to(dev).train()
    fl = synthetic_tensor((1, 32, h4, w4), 5).to(dev).to(low).requires_grad_()
    fr = synthetic_tensor((1, 32, h4, w4), 6).to(dev).to(low).requires_grad_()
    with env.ctx():
        with torch.amp.autocast(dev.type, dtype=low):
            under = m.aggregate(fl, fr, 4 * h4, 4 * w4)
        sum(p.sum() for p in under).backward()
        with torch.no_grad():
            plain = m.aggregate(fl.detach().float(), fr.detach().float(), 4 * h4, 4 * w4)
    assert len(under) == 3
    for a, b in zip(under, plain):
        assert a.dtype == torch.float32 and torch.equal(a.detach(), b)
    assert fl.grad is not None and fl.grad.dtype == low and torch.isfinite(fl.grad.float()).all()
    assert fr.grad is not None and fr.grad.dtype == low


def test_functional_api_under_autocast(env):
    """The drop-in functions (models/GwcNet/submodule.py:30-63, disparity_estimators) with low-precision operands under
    autocast: fp32 results equal to the fp32 call on the cast-up operands; outside autocast the tensors are refused."""
    from stereo_toolbox_amd._capi import StxError
    from stereo_toolbox_amd.disparity_estimators import softargmax_disparity_estimator
    from stereo_toolbox_amd.models.GwcNet.submodule import build_concat_volume, build_gwc_volume
    dev = env.device
    low = torch.float16 if dev.type == "cuda" else torch.bfloat16
    a, b = synthetic_tensor((2, 16, 5, 11), 7).to(dev).to(low), synthetic_tensor((2, 16, 5, 11), 8).to(dev).to(low)
    x = torch.softmax(synthetic_tensor((2, 16, 6, 10), 9) * 3, 1).to(dev).to(low)
    with env.ctx():
        with torch.amp.autocast(dev.type, dtype=low):
            g, c, s = build_gwc_volume(a, b, 6, 4), build_concat_volume(a, b, 6), softargmax_disparity_estimator(x, 16)
        g32, c32, s32 = build_gwc_volume(a.float(), b.float(), 6, 4), build_concat_volume(a.float(), b.float(), 6), \
            softargmax_disparity_estimator(x.float(), 16)
        for u, v in ((g, g32), (c, c32), (s, s32)):
            assert u.dtype == torch.float32 and torch.equal(u, v)
        if env.name == "hip":                                   # (the emulator hook replaces _chk by an assert)
            with pytest.raises(StxError, match="float32"):
                build_gwc_volume(a, b, 6, 4)
