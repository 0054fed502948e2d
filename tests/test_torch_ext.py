"""The TORCH_LIBRARY loader (north_star: kernels "loaded via torch.utils.cpp_extension"; SURVEY.md 8(b) "C++/HIP extension
ABI"): `stereo_toolbox_amd.torch_ext.load()` builds csrc/torch_binding.cpp in-tree against libstx_hip.so and registers
`torch.ops.stx.*`.  CPU: it builds, the schemas exist, shape inference runs on the Meta backend (FakeTensor / torch.compile
tracing), CPU tensors are refused (no fallback).  GPU: the operators give bit for bit what the ctypes binding gives (same
kernels, same library), and the volume builder differentiates through the dispatcher."""
import pytest
import torch


@pytest.fixture(scope="module")
def stx():
    import stereo_toolbox_amd.torch_ext as tx
    return tx.load()


def test_ops_registered_with_schemas(stx):
    names = ["cost_volume", "conv3d_pack_weight", "conv3d", "deconv3d", "conv3d_wgrad", "regression_head", "softargmax",
             "argmax_disparity", "build_info"]
    for n in names:
        assert hasattr(stx, n), n
    s = str(torch.ops.stx.cost_volume.default._schema)
    assert "Tensor? Lg" in s and "int maxdisp" in s and "bool mask_left" in s and "-> Tensor" in s
    assert "gfx950" in stx.build_info()                   # the binding is linked against the gfx950 library, not the emulator


def test_meta_kernels_infer_shapes(stx):
    m = torch.device("meta")
    Lg, Lc = torch.empty(2, 320, 16, 32, device=m), torch.empty(2, 12, 16, 32, device=m)
    vol = stx.cost_volume(Lg, Lg, Lc, Lc, 16, 40, True)
    assert vol.shape == (2, 16, 16, 32, 64) and vol.device.type == "meta"
    assert stx.cost_volume(None, None, Lc, Lc, 16, 0, False).shape == (2, 16, 16, 32, 24)
    x = torch.empty(1, 8, 12, 20, 32, device=m)
    wp = stx.conv3d_pack_weight(torch.empty(64, 32, 3, 3, 3, device=m), 0)
    assert wp.shape == (27 * 4 * 2 * 256,)
    assert stx.conv3d(x, wp, 64, 3, 2, None, None, None, 1).shape == (1, 4, 6, 10, 64)
    assert stx.deconv3d(x, wp, 16, None, None, None, 0).shape == (1, 16, 24, 40, 16)
    assert stx.conv3d_wgrad(x, torch.empty(1, 8, 12, 20, 64, device=m), 3, 1).shape == (64, 32, 27)
    assert stx.regression_head(torch.empty(2, 12, 16, 32, device=m), 48, 64, 128, False).shape == (2, 64, 128)
    assert stx.softargmax(torch.empty(2, 48, 8, 9, device=m)).shape == (2, 1, 8, 9)
    assert stx.argmax_disparity(torch.empty(2, 48, 8, 9, device=m)).dtype == torch.int64


def test_cpu_tensors_are_refused(stx):
    with pytest.raises((NotImplementedError, RuntimeError)):
        stx.softargmax(torch.zeros(1, 4, 2, 2))
    with pytest.raises((NotImplementedError, RuntimeError)):
        stx.cost_volume(torch.zeros(1, 8, 2, 16), torch.zeros(1, 8, 2, 16), None, None, 4, 2, True)


@pytest.mark.gpu
def test_ops_match_the_ctypes_binding(stx):
    from stereo_toolbox_amd import ops
    from stereo_toolbox_amd.utils import synthetic_tensor
    dev = torch.device("cuda:0")
    Lg, Rg = synthetic_tensor((1, 320, 6, 40), 1).to(dev), synthetic_tensor((1, 320, 6, 40), 2).to(dev)
    Lc, Rc = synthetic_tensor((1, 12, 6, 40), 3).to(dev), synthetic_tensor((1, 12, 6, 40), 4).to(dev)
    assert torch.equal(stx.cost_volume(Lg, Rg, Lc, Rc, 16, 40, True), ops.cost_volume_forward(Lg, Rg, Lc, Rc, 16, 40, True))
    x = synthetic_tensor((1, 4, 6, 40, 32), 5).to(dev)
    w = (synthetic_tensor((32, 32, 3, 3, 3), 6) * 0.2).to(dev)
    wp = stx.conv3d_pack_weight(w, 0)
    assert torch.equal(wp, ops.pack_weight(w, 0))
    sc, bs = synthetic_tensor((32,), 7, lo=0.5, hi=1.5).to(dev), synthetic_tensor((32,), 8).to(dev)
    assert torch.equal(stx.conv3d(x, wp, 32, 3, 1, sc, bs, None, 1), ops.conv3d_forward(x, wp, 32, 3, 1, sc, bs, None, True)[0])
    gy = synthetic_tensor((1, 4, 6, 40, 32), 9).to(dev)
    assert torch.equal(stx.conv3d_wgrad(x, gy, 3, 1), ops.conv3d_wgrad(x, gy, 3, 1))
    cost = (synthetic_tensor((1, 4, 6, 40), 10) * 3).to(dev)
    assert torch.equal(stx.regression_head(cost, 16, 24, 160, False), ops.regression_head(cost, 16, 24, 160))
    p = torch.softmax(synthetic_tensor((2, 16, 6, 10), 11).to(dev) * 3, 1)
    assert torch.equal(stx.softargmax(p), ops.softargmax(p, 16, keepdim=True))
    assert torch.equal(stx.argmax_disparity(p), ops.argmax_disparity(p))


@pytest.mark.gpu
def test_cost_volume_autograd_through_the_dispatcher(stx):
    from stereo_toolbox_amd import ops
    from stereo_toolbox_amd.utils import synthetic_tensor
    dev = torch.device("cuda:0")
    leaves = [synthetic_tensor(s, i).to(dev).requires_grad_() for i, s in enumerate([(1, 64, 5, 37)] * 2 + [(1, 4, 5, 37)] * 2)]
    twins = [t.detach().clone().requires_grad_() for t in leaves]
    g = synthetic_tensor((1, 12, 5, 37, 16), 9).to(dev)
    stx.cost_volume(*leaves, 12, 8, True).backward(g)
    ops.cost_volume(*twins, 12, 8, mask_left=True).backward(g)
    for a, b in zip(leaves, twins):
        assert a.grad is not None and torch.equal(a.grad, b.grad)
