"""Diagnostic (GPU): where do the blocked-sum variants of conv3d_marchw_kernel differ from the sequential kernel?"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_toolbox_amd._capi import get_lib
lib = get_lib()
dev = torch.device("cuda:0")
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def conv(x, w, bs):
    lib.set_tuning("STX_MARCH_BS", bs)
    B, D, H, W, Cin = x.shape
    Cout = w.shape[0]
    wp = torch.empty(lib.raw("stx_conv3d_packed_floats")(Cin, Cout, 27), device=dev)
    lib.call("stx_conv3d_pack_weight", P(w), P(wp), Cout, Cin, 27, 0, st())
    out = torch.full((B, D, H, W, Cout), float("nan"), device=dev)
    lib.call("stx_conv3d_fwd", P(x), P(wp), P(out), None, None, None, None, B, D, H, W, Cin, Cout, 3, 1, 0, st())
    torch.cuda.synchronize()
    return out


for shape in ((1, 5, 9, 37), (1, 12, 16, 32)):
    B, D, H, W = shape
    for kind in ("ones", "randn"):
        torch.manual_seed(0)
        x = torch.ones(B, D, H, W, 32, device=dev) if kind == "ones" else torch.randn(B, D, H, W, 32, device=dev)
        w = torch.ones(32, 32, 3, 3, 3, device=dev) if kind == "ones" else torch.randn(32, 32, 3, 3, 3, device=dev) * 0.1
        ref = conv(x, w, 0)
        for bs in (1,):
            got = conv(x, w, bs)
            diff = (got - ref)
            bad = diff.abs() > 1e-3 * ref.abs().max()
            print(f"shape {shape} {kind} BS={bs}: bad {int(bad.sum())} of {bad.numel()}, max |diff| {float(diff.abs().max()):.4g}, "
                  f"nan {int(torch.isnan(got).sum())}")
            if bad.any():
                idx = bad.nonzero()
                for dim, name in ((1, "d"), (2, "h"), (3, "w"), (4, "c")):
                    vals, counts = idx[:, dim].unique(return_counts=True)
                    print(f"   bad by {name}: " + " ".join(f"{int(v)}:{int(c)}" for v, c in zip(vals, counts))[:300])
                if kind == "ones":
                    dv, dc = diff[bad].unique(return_counts=True)
                    print("   diff values: " + " ".join(f"{float(v):.0f}x{int(c)}" for v, c in zip(dv, dc))[:300])
