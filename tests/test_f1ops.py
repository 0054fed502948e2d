"""2-D helpers of the PCWNet / CFNet family on the HIP kernels of csrc/refine2d.hip: `warp`,
`build_corrleation_volume`, `disparity_variance`, `disparity_variance_confidence` -- forward and backward through the C-ABI
(emulator on the CPU, gfx950 library on the GPU) against fixtures produced by the REFERENCE's own functions
(tests/golden/make_golden_f1ops.py) and against the oracle's restatements on further shapes."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests.backends import BACKENDS, Backend  # noqa: F401
from tests.golden.f1ops_config import CORR_CASES, VAR_CASES, WARP_CASES, corr_inputs, var_inputs, warp_inputs
from stereo_toolbox_amd.utils import synthetic_tensor

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f1ops.npz"))


@pytest.fixture(params=BACKENDS)
def be(request):
    return Backend(request.param)


def _g(name):
    return torch.from_numpy(GOLD[name])


def _close(a, b, tol, what):
    err = (a.cpu() - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), (what, err)


def test_oracle_matches_reference_fixtures():
    """The oracle's restatements (pcw_warp, pcw_correlation_volume, cf_disparity_variance*) against the reference's outputs and
    gradients -- the pin of the checker itself."""
    for k, (name, B, C, H, W, lo, hi) in enumerate(WARP_CASES):
        x, d, g = warp_inputs(B, C, H, W, lo, hi, 100 + 10 * k)
        x.requires_grad_(); d.requires_grad_()
        y = O.pcw_warp(x, d)
        y.backward(g)
        assert torch.equal(y.detach(), _g(f"warp_{name}"))
        assert torch.equal(x.grad, _g(f"warp_{name}_gx")) and torch.equal(d.grad, _g(f"warp_{name}_gd"))
    for k, (name, B, C, H, W, md, G) in enumerate(CORR_CASES):
        a, b, g = corr_inputs(B, C, H, W, md, G, 200 + 10 * k)
        a.requires_grad_(); b.requires_grad_()
        v = O.pcw_correlation_volume(a, b, md, G)
        v.backward(g)
        assert torch.equal(v.detach(), _g(f"corr_{name}"))
        _close(a.grad, _g(f"corr_{name}_ga"), 1e-6, "ga"); _close(b.grad, _g(f"corr_{name}_gb"), 1e-6, "gb")
    for k, (name, B, D, H, W) in enumerate(VAR_CASES):
        x, d, s, g = var_inputs(B, D, H, W, 300 + 10 * k)
        assert torch.equal(O.cf_disparity_variance(x, D, d), _g(f"var_{name}"))
        assert torch.equal(O.cf_disparity_variance_confidence(x, s, d), _g(f"conf_{name}"))


@pytest.mark.parametrize("case", range(len(WARP_CASES)))
def test_warp_fwd_bwd(be, case):
    name, B, C, H, W, lo, hi = WARP_CASES[case]
    x, d, g = warp_inputs(B, C, H, W, lo, hi, 100 + 10 * case)
    out, gx, gd = be.empty(B, C, H, W), be.empty(B, C, H, W), be.empty(B, 1, H, W)
    be.call("stx_warp_fwd", be.dev(x), be.dev(d), out, B, C, H, W)
    _close(out, _g(f"warp_{name}"), 2e-6, "warp")
    # the validity mask is a discontinuity: exactly the reference's pixels are zeroed
    assert torch.equal(out.cpu() == 0, _g(f"warp_{name}") == 0)
    be.call("stx_warp_bwd", be.dev(g), be.dev(x), be.dev(d), gx, gd, B, C, H, W)
    _close(gx, _g(f"warp_{name}_gx"), 2e-6, "gx")
    _close(gd, _g(f"warp_{name}_gd"), 1e-5, "gdisp")
    # either gradient alone
    gx2, gd2 = be.empty(B, C, H, W), be.empty(B, 1, H, W)
    be.call("stx_warp_bwd", be.dev(g), be.dev(x), be.dev(d), gx2, None, B, C, H, W)
    be.call("stx_warp_bwd", be.dev(g), be.dev(x), be.dev(d), None, gd2, B, C, H, W)
    _close(gx2, _g(f"warp_{name}_gx"), 2e-6, "gx alone")
    assert torch.equal(gd2, gd)


def test_warp_edge_cases(be):
    """Huge disparities sample nothing (zero output, zero gradients, no out-of-range access); a NaN disparity touches no memory
    and PROPAGATES: NaN output at that pixel and a NaN disparity gradient, as the reference's grid_sample / mask logic gives
    (PCWNet/submodule.py:166-176) -- a diverged refinement must show as a NaN loss; a 1-column image."""
    B, C, H, W = 1, 2, 4, 9
    x = synthetic_tensor((B, C, H, W), 5)
    d = synthetic_tensor((B, 1, H, W), 6, lo=0.0, hi=3.0)
    d[0, 0, 1, 2], d[0, 0, 2, 3], d[0, 0, 2, 4] = float("nan"), 1e30, -1e30
    out = be.empty(B, C, H, W)
    be.call("stx_warp_fwd", be.dev(x), be.dev(d), out, B, C, H, W)
    ref = O.pcw_warp(x, d)
    assert torch.isnan(ref[0, :, 1, 2]).all() and torch.isnan(out.cpu()[0, :, 1, 2]).all()          # the NaN disparity's pixel, both sides
    assert torch.equal(torch.isnan(out.cpu()), torch.isnan(ref))
    _close(torch.nan_to_num(out.cpu()), torch.nan_to_num(ref), 2e-6, "warp")
    assert out[0, :, 2, 3:5].abs().max().item() == 0
    gx, gd = be.empty(B, C, H, W), be.empty(B, 1, H, W)
    be.call("stx_warp_bwd", be.dev(torch.ones(B, C, H, W)), be.dev(x), be.dev(d), gx, gd, B, C, H, W)
    assert torch.isfinite(gx).all() and torch.isnan(gd.cpu()[0, 0, 1, 2]) and gd[0, 0, 2, 3].item() == 0
    assert torch.isfinite(torch.cat((gd.cpu().flatten()[:11], gd.cpu().flatten()[12:]))).all()
    x1, d1 = synthetic_tensor((1, 1, 3, 1), 7), torch.zeros(1, 1, 3, 1)
    o1 = be.empty(1, 1, 3, 1)
    be.call("stx_warp_fwd", be.dev(x1), be.dev(d1), o1, 1, 1, 3, 1)
    _close(o1, O.pcw_warp(x1, d1), 2e-6, "1 column")


@pytest.mark.parametrize("case", range(len(CORR_CASES)))
def test_corr_volume_fwd_bwd(be, case):
    name, B, C, H, W, md, G = CORR_CASES[case]
    a, b, g = corr_inputs(B, C, H, W, md, G, 200 + 10 * case)
    vol = be.empty(B, G, 2 * md + 1, H, W)
    be.call("stx_corr_volume_fwd", be.dev(a), be.dev(b), vol, B, C, H, W, md, G)
    ref = _g(f"corr_{name}")
    _close(vol, ref, 2e-6, "corr")
    assert torch.equal(vol.cpu() == 0, ref == 0)          # the zero regions (w < i, columns >= |i| of the negative slices) exactly
    ga, gb = be.empty(B, C, H, W), be.empty(B, C, H, W)
    be.call("stx_corr_volume_bwd", be.dev(g), be.dev(a), be.dev(b), ga, gb, B, C, H, W, md, G)
    _close(ga, _g(f"corr_{name}_ga"), 3e-6, "gref")
    _close(gb, _g(f"corr_{name}_gb"), 3e-6, "gtgt")
    ga2, gb2 = be.empty(B, C, H, W), be.empty(B, C, H, W)
    be.call("stx_corr_volume_bwd", be.dev(g), be.dev(a), be.dev(b), ga2, None, B, C, H, W, md, G)
    be.call("stx_corr_volume_bwd", be.dev(g), be.dev(a), be.dev(b), None, gb2, B, C, H, W, md, G)
    assert torch.equal(ga2, ga) and torch.equal(gb2, gb)


@pytest.mark.parametrize("shape", [(1, 40, 2, 24, 24, 1), (1, 6, 1, 300, 48, 2), (2, 3, 3, 5, 0, 3), (1, 70, 2, 131, 3, 1)])
def test_corr_volume_vs_oracle(be, shape):
    """Further shapes against the oracle: maxdisp == W, the 48-disparity instantiation over three column tiles, maxdisp 0,
    more channels per group than one staging pass holds."""
    B, C, H, W, md, G = shape
    a, b, g = corr_inputs(B, C, H, W, md, G, 400 + C)
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    ref = O.pcw_correlation_volume(ar, br, md, G)
    ref.backward(g)
    vol, ga, gb = be.empty(B, G, 2 * md + 1, H, W), be.empty(B, C, H, W), be.empty(B, C, H, W)
    be.call("stx_corr_volume_fwd", be.dev(a), be.dev(b), vol, B, C, H, W, md, G)
    be.call("stx_corr_volume_bwd", be.dev(g), be.dev(a), be.dev(b), ga, gb, B, C, H, W, md, G)
    _close(vol, ref.detach(), 3e-6, "corr")
    _close(ga, ar.grad, 4e-6, "gref")
    _close(gb, br.grad, 4e-6, "gtgt")


def test_corr_volume_rejects_bad_arguments(be):
    from stereo_toolbox_amd._capi import StxError
    a = be.dev(synthetic_tensor((1, 6, 2, 20), 1))
    vol = be.empty(1, 4, 9, 2, 20)
    with pytest.raises(StxError):
        be.call("stx_corr_volume_fwd", a, a, vol, 1, 6, 2, 20, 4, 4)       # 6 channels, 4 groups (reference: assert)
    with pytest.raises(StxError):
        be.call("stx_corr_volume_fwd", a, a, vol, 1, 6, 2, 20, 21, 1)      # maxdisp > W: the reference's slices mismatch too
    with pytest.raises(StxError):
        be.call("stx_corr_volume_fwd", a, a, vol, 1, 6, 2, 20, 49, 1)


@pytest.mark.parametrize("case", range(len(VAR_CASES)))
def test_disparity_variance_fwd_bwd(be, case):
    name, B, D, H, W = VAR_CASES[case]
    x, d, s, g = var_inputs(B, D, H, W, 300 + 10 * case)
    out = be.empty(B, 1, H, W)
    be.call("stx_disparity_variance_fwd", be.dev(x), be.dev(d), None, out, B, D, H * W)
    _close(out, _g(f"var_{name}"), 1e-6, "variance")
    gx, gd = be.empty(B, D, H, W), be.empty(B, 1, H, W)
    be.call("stx_disparity_variance_bwd", be.dev(g), be.dev(x), be.dev(d), None, gx, gd, None, B, D, H * W)
    _close(gx, _g(f"var_{name}_gx"), 1e-6, "gx")
    _close(gd, _g(f"var_{name}_gd"), 2e-6, "gdisp")
    be.call("stx_disparity_variance_fwd", be.dev(x), be.dev(d), be.dev(s), out, B, D, H * W)
    _close(out, _g(f"conf_{name}"), 1e-6, "confidence")
    gs = be.empty(B, D, H, W)
    be.call("stx_disparity_variance_bwd", be.dev(g), be.dev(x), be.dev(d), be.dev(s), gx, gd, gs, B, D, H * W)
    _close(gx, _g(f"conf_{name}_gx"), 1e-6, "gx")
    _close(gd, _g(f"conf_{name}_gd"), 2e-6, "gdisp")
    _close(gs, _g(f"conf_{name}_gs"), 2e-6, "gsamples")


@pytest.mark.parametrize("envname", ["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def test_functional_api_autograd(envname):
    """The drop-in functions (`models.PCWNet.submodule.warp / build_corrleation_volume`, `models.CFNet.submodule.
    disparity_variance / _confidence`) through autograd, chained the way PCWNet's refinement uses them (pcwnet.py:465-469):
    values and all input gradients against the oracle's chain."""
    import contextlib
    from stereo_toolbox_amd.models.CFNet.submodule import disparity_variance, disparity_variance_confidence
    from stereo_toolbox_amd.models.PCWNet.submodule import build_corrleation_volume, warp
    if envname == "hip" and not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    dev = torch.device("cuda:0" if envname == "hip" else "cpu")
    if envname == "emu":
        from tests.emu_util import emu_product_path
        ctx = emu_product_path()
    else:
        ctx = contextlib.nullcontext()
    B, C, H, W, md = 1, 8, 6, 60, 24
    left, right = synthetic_tensor((B, C, H, W), 11), synthetic_tensor((B, C, H, W), 12)
    disp = synthetic_tensor((B, 1, H, W), 13, lo=0.0, hi=20.0)
    g = synthetic_tensor((B, C + 2 * md + 1, H, W), 14)

    def chain(warp_fn, corr_fn, l_, r_, d_):
        rw = warp_fn(r_, d_)
        return torch.cat((l_ - rw, corr_fn(l_, rw, md, 1).squeeze(1)), dim=1)
    ref_in = [t.clone().requires_grad_() for t in (left, right, disp)]
    ref = chain(O.pcw_warp, O.pcw_correlation_volume, *ref_in)
    ref.backward(g)
    got_in = [t.clone().to(dev).requires_grad_() for t in (left, right, disp)]
    with ctx:
        got = chain(warp, build_corrleation_volume, *got_in)
        got.backward(g.to(dev))
    _close(got.detach(), ref.detach(), 3e-6, "chain")
    for a, b, n in zip(got_in, ref_in, ("left", "right", "disp")):
        _close(a.grad, b.grad, 1e-5, n)
    # variance pair
    x, d, s, gv = var_inputs(2, 12, 5, 7, 900)
    rin = [t.clone().requires_grad_() for t in (x, d, s)]
    (O.cf_disparity_variance(rin[0], 12, rin[1]) + O.cf_disparity_variance_confidence(rin[0], rin[2], rin[1])).backward(gv)
    gin = [t.clone().to(dev).requires_grad_() for t in (x, d, s)]
    with ctx:
        v = disparity_variance(gin[0], 12, gin[1]) + disparity_variance_confidence(gin[0], gin[2], gin[1])
        v.backward(gv.to(dev))
    for a, b, n in zip(gin, rin, ("x", "disparity", "samples")):
        _close(a.grad, b.grad, 3e-6, n)
    with pytest.raises(AssertionError):
        with ctx:
            build_corrleation_volume(got_in[0].detach(), got_in[1].detach(), md, 3)      # 8 channels, 3 groups
