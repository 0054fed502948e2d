"""Per-kernel timing of the C-ABI entry points at the GwcNet_GC 576x960, D=192 shapes (MI355X only).

python tools/kernel_bench.py [--iters N] [--only substr] -> one JSON line per kernel with ms,
TFLOP/s (true MACs x2) or GB/s (algorithmic bytes).  Used for profiling; bench.py is the contract.
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_toolbox_amd import _capi  # noqa: E402
if os.environ.get("STX_BENCH_LIB"):          # A/B a variant build of the library (tuning only)
    _capi.LIB_PATH = os.path.abspath(os.environ["STX_BENCH_LIB"])
from stereo_toolbox_amd._capi import get_lib  # noqa: E402

lib = get_lib()
dev = torch.device("cuda:0")


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


COLD = False
_FLUSH = None


def timeit(fn, iters):
    """Mean launch time.  Default: back-to-back launches (operands up to the 256 MiB Infinity Cache stay resident between
    iterations: streaming kernels then report more than the HBM rate).  --cold: a 512 MiB fill between iterations evicts the
    operands from L2 and the Infinity Cache, every launch is timed by its own event pair -- the number a kernel sees inside
    a train step, where hundreds of MB of other tensors pass between two uses of anything."""
    global _FLUSH
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    if not COLD:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    if _FLUSH is None:
        _FLUSH = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)
    evs = []
    for i in range(iters):
        _FLUSH.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / iters


def report(name, ms, flops=None, nbytes=None):
    r = {"kernel": name, "ms": round(ms, 4), "cache": "cold" if COLD else "warm"}
    if flops:
        r["TFLOPs"] = round(flops / ms / 1e9, 2)
        r["mfma_frac"] = round(flops / ms / 1e9 / 157.3, 3)
    if nbytes:
        r["GBs"] = round(nbytes / ms / 1e6, 1)
        r["hbm_frac"] = round(nbytes / ms / 1e6 / 8000.0, 3)
    print(json.dumps(r), flush=True)


def pack(w, mode):
    A, Bd = w.shape[0], w.shape[1]
    T = w[0, 0].numel()
    K, N = (Bd, A) if mode == 0 else (A, Bd)
    wp = torch.empty(lib.raw("stx_conv3d_packed_floats")(K, N, T), device=dev)
    lib.call("stx_conv3d_pack_weight", P(w), P(wp), A, Bd, T, mode, stream())
    return wp


# A/B switches of libstx_hip.so (StxTune in csrc/stx_common.h), flipped inside one process through stx_set_tuning: a whole
# comparison costs one interpreter start.  (label, kernel filter, {switch: value})
AB_SETS = [
    ("march kernel: one sequential accumulation chain per output", "conv_32_32_L0_fwd,conv_64_32_L0_fwd", {"STX_MARCH_BS": 0}),
    ("march kernel: general (branchy) epilogue for every launch", "conv_32_32_L0_fwd,conv_64_32_L0_fwd", {"STX_MARCH_EPI": 0}),
    ("march kernel: no plane staging (ablation)", "conv_32_32_L0_fwd", {"STX_MARCH_ABLATE": 1}),
    ("march kernel: no epilogue stores (ablation)", "conv_32_32_L0_fwd", {"STX_MARCH_ABLATE": 2}),
    ("64->64 L1 on the march kernel (2 x 2 channel slices)", "conv_64_64_L1_fwd", {"STX_CONV_L1_MARCH": 1}),
    ("implicit GEMM: one row x all column blocks per wave (the wave grid of rounds 1-4)", "conv_64_64_L1_fwd,conv_32_64_s2_L0_fwd,conv_128_128_L2_fwd,conv_64_128_s2_L1_fwd", {"STX_CONV_WN": 1}),
    ("implicit GEMM: 128 output channels as two rows x two blocks per wave", "conv_128_128_L2_fwd,conv_64_128_s2_L1_fwd", {"STX_CONV_WN": 3}),
    ("implicit GEMM: 128 output channels as four rows x one block per wave", "conv_128_128_L2_fwd,conv_64_128_s2_L1_fwd", {"STX_CONV_WN": 4}),
    ("stride-2 32->64 with the padded LDS tile", "conv_32_64_s2_L0_fwd", {"STX_CONV_S2_DENSE": 0}),
    ("weight gradient 3x3x3 s1: tile kernel of rounds 1-3 instead of the march kernel", "conv_32_32_L0_wgrad,conv_64_32_L0_wgrad,conv_64_64_L1_wgrad,conv_128_128_L2_wgrad", {"STX_WGRAD_MARCH": 0}),
    ("weight gradient 3x3x3 s2 / transposed: tile kernel of rounds 1-3 instead of the parity-split march kernel", "conv_32_64_s2_L0_wgrad,conv_64_128_s2_L1_wgrad", {"STX_WGRAD_MARCH": 1}),
    ("s2 march weight gradient: no staging loads (ablation)", "conv_32_64_s2_L0_wgrad,conv_64_128_s2_L1_wgrad", {"STX_WGRAD_ABLATE": 1}),
    ("s2 march weight gradient: no MFMA groups (ablation)", "conv_32_64_s2_L0_wgrad,conv_64_128_s2_L1_wgrad", {"STX_WGRAD_ABLATE": 2}),
    ("s2 march weight gradient: no LDS writes (ablation)", "conv_32_64_s2_L0_wgrad,conv_64_128_s2_L1_wgrad", {"STX_WGRAD_ABLATE": 3}),
    ("weight gradient: no tile staging (ablation)", "conv_32_32_L0_wgrad,conv_64_64_L1_wgrad,conv_32_64_s2_L0_wgrad", {"STX_WGRAD_ABLATE": 1}),
    ("weight gradient: no MFMA loop (ablation)", "conv_32_32_L0_wgrad,conv_64_64_L1_wgrad,conv_32_64_s2_L0_wgrad", {"STX_WGRAD_ABLATE": 2}),
    ("ACVNet patch convolution: cache-fed kernel of rounds 3-4 instead of the rolling window", "dwconv_hw", {"STX_DWCONV_ROLL": 0}),
    ("sampled volume bwd: global atomics only (first version)", "sampled_volume", {"STX_SV_BWD_V1": 1}),
    ("cost volume fwd: first-generation fallback kernel", "cost_volume_fwd", {"STX_CV_OLD": 1}),
    ("cost volume fwd: cache-line pairs through the LDS-DMA slot", "cost_volume_fwd", {"STX_CV_PF": 2}),
    ("cost volume fwd: runs cut at whole macro-units", "cost_volume_fwd", {"STX_CV_UNITS": 0}),
    ("cost volume fwd: windows of 256 macro-units", "cost_volume_fwd", {"STX_CV_WIN": 256}),
    ("cost volume fwd: windows of 384 macro-units", "cost_volume_fwd", {"STX_CV_WIN": 384}),
    ("cost volume fwd: windows of 512 macro-units", "cost_volume_fwd", {"STX_CV_WIN": 512}),
    ("cost volume fwd: windows of 720 macro-units", "cost_volume_fwd", {"STX_CV_WIN": 720}),
    ("cost volume fwd: windows of 1080 macro-units", "cost_volume_fwd", {"STX_CV_WIN": 1080}),
    ("cost volume bwd: first-generation fallback kernel", "cost_volume_bwd", {"STX_CVB_OLD": 1}),
    ("cost volume bwd: team schedule", "cost_volume_bwd", {"STX_CVB_TEAM": 1}),
    ("cost volume bwd: 2 chunk sets in flight", "cost_volume_bwd", {"STX_CVB_NSET": 2}),
    ("cost volume bwd: 4 chunk sets in flight", "cost_volume_bwd", {"STX_CVB_NSET": 4}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ab", action="store_true", help="after the table: re-time the kernels of AB_SETS under their switch")
    ap.add_argument("--ab-filter", default="", help="substring filter on the A/B set labels")
    ap.add_argument("--cold", action="store_true", help="cold-cache timing: 512 MiB fill between iterations (see timeit)")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="", help="substring filter(s) on kernel names, comma separated")
    ap.add_argument("--skip-wgrad", action="store_true")
    ap.add_argument("--H", type=int, default=576)
    ap.add_argument("--W", type=int, default=960)
    ap.add_argument("--D", type=int, default=192)
    a = ap.parse_args()
    global COLD
    COLD = a.cold
    run_table(a, a.only)
    if a.ab:
        for label, flt, env in AB_SETS:
            if a.ab_filter and a.ab_filter not in label:
                continue
            print(json.dumps({"ab": label, "tuning": env}), flush=True)
            old = {k: lib.set_tuning(k, v) for k, v in env.items()}
            try:
                run_table(a, flt, only_exact=True)
            finally:
                for k, v in old.items():
                    lib.set_tuning(k, v)


def run_table(a, only_arg, only_exact=False):
    B, H4, W4, D4 = 1, a.H // 4, a.W // 4, a.D // 4
    L = {0: (D4, H4, W4), 1: (D4 // 2, H4 // 2, W4 // 2), 2: (D4 // 4, H4 // 4, W4 // 4)}
    it = a.iters

    only = [t for t in only_arg.split(",") if t]

    def want(n):
        # a section asks with its prefix ("cost_volume", "head") or a full kernel name; a filter matches either way
        return not only or any(t in n or (only_exact and n in t) for t in only)

    if want("cost_volume"):
        Lg, Rg = torch.randn(B, 320, H4, W4, device=dev), torch.randn(B, 320, H4, W4, device=dev)
        Lc, Rc = torch.randn(B, 12, H4, W4, device=dev), torch.randn(B, 12, H4, W4, device=dev)
        vol = torch.empty(B, D4, H4, W4, 64, device=dev)
        nb = (Lg.numel() * 2 + Lc.numel() * 2 + vol.numel()) * 4
        ms = timeit(lambda: lib.call("stx_cost_volume_fwd", P(Lg), P(Rg), 320, 40, P(Lc), P(Rc), 12, None, P(vol),
                                     B, H4, W4, D4, 1, stream()), it)
        report("cost_volume_fwd_gwcgc", ms, nbytes=nb)
        g = [torch.empty_like(t) for t in (Lg, Rg, Lc, Rc)]
        ms = timeit(lambda: lib.call("stx_cost_volume_bwd", P(vol), P(Lg), P(Rg), 320, 40, 12, P(g[0]), P(g[1]), P(g[2]),
                                     P(g[3]), B, H4, W4, D4, 1, stream()), it)
        report("cost_volume_bwd_gwcgc", ms, nbytes=nb + Lg.numel() * 8)
        Lp, Rp = torch.randn(B, 32, H4, W4, device=dev), torch.randn(B, 32, H4, W4, device=dev)
        ms = timeit(lambda: lib.call("stx_cost_volume_fwd", None, None, 0, 0, P(Lp), P(Rp), 32, None, P(vol), B, H4, W4,
                                     D4, 1, stream()), it)
        report("cost_volume_fwd_psm_concat", ms, nbytes=(Lp.numel() * 2 + vol.numel()) * 4)
        del Lg, Rg, vol, g

    if want("conv2d"):
        # the 2-D feature CNN's 3x3 stride-1 layers on csrc/conv2d.hip (both views batched): forward (with the BatchNorm statistics
        # epilogue, two views) and data gradient
        for name, Ci, Co, Hh, Ww in (("conv2d_64_64_quarter", 64, 64, a.H // 4, a.W // 4), ("conv2d_32_32_half", 32, 32, a.H // 2, a.W // 2)):
            x2 = torch.randn(2, Hh, Ww, Ci, device=dev)
            w2 = torch.randn(Co, 3, 3, Ci, device=dev) * 0.05
            o2 = torch.empty(2, Hh, Ww, Co, device=dev)
            st2 = torch.empty(int(lib.raw("stx_conv2d_stat_rows")(2)), 2, Co, device=dev)
            fl = 2.0 * 2 * Hh * Ww * Ci * Co * 9
            ms = timeit(lambda: lib.call("stx_conv2d_fwd", P(x2), P(w2), P(o2), P(st2), 2, Hh, Ww, Ci, Co, 0, 2, stream()), it)
            report(name + "_fwd", ms, flops=fl)
            ms = timeit(lambda: lib.call("stx_conv2d_fwd", P(o2), P(w2), P(x2), None, 2, Hh, Ww, Co, Ci, 1, 1, stream()), it)
            report(name + "_dgrad", ms, flops=fl)
        del x2, w2, o2, st2

    if want("sampled_volume"):
        # CFNet cascade stage at 1/4 resolution: 40 groups x 4 channels + 12 concat channels, 16 hypotheses, 72-channel voxels
        S, G, cpg, Cc, CTp = 16, 40, 4, 12, 72
        Lg, Rg = torch.randn(B, G * cpg, H4, W4, device=dev), torch.randn(B, G * cpg, H4, W4, device=dev)
        Lc, Rc = torch.randn(B, Cc, H4, W4, device=dev), torch.randn(B, Cc, H4, W4, device=dev)
        base = torch.rand(B, 1, H4, W4, device=dev) * 30
        smp = (base + torch.arange(S, device=dev).view(1, S, 1, 1)).floor()
        vol = torch.empty(B, S, H4, W4, CTp, device=dev)
        nb = (Lg.numel() * 2 + Lc.numel() * 2 + smp.numel() + vol.numel()) * 4
        ms = timeit(lambda: lib.call("stx_sampled_volume_fwd", P(Lg), P(Rg), G * cpg, G, P(Lc), P(Rc), Cc, P(smp), P(vol),
                                     B, H4, W4, S, CTp, stream()), it)
        report("sampled_volume_fwd_cfnet_s3", ms, nbytes=nb)
        g = [torch.empty_like(t) for t in (Lg, Rg, Lc, Rc)]
        ms = timeit(lambda: lib.call("stx_sampled_volume_bwd", P(vol), P(Lg), P(Rg), G * cpg, G, Cc, P(smp), P(g[0]), P(g[1]),
                                     P(g[2]), P(g[3]), B, H4, W4, S, CTp, stream()), it)
        report("sampled_volume_bwd_cfnet_s3", ms, nbytes=nb + Lg.numel() * 8)
        del Lg, Rg, vol, g

    convs = [  # name, level_in, Cin, Cout, ks, stride
        ("conv_64_32_L0", 0, 64, 32, 3, 1), ("conv_32_32_L0", 0, 32, 32, 3, 1), ("conv_32_64_s2_L0", 0, 32, 64, 3, 2),
        ("conv_64_64_L1", 1, 64, 64, 3, 1), ("conv_64_128_s2_L1", 1, 64, 128, 3, 2), ("conv_128_128_L2", 2, 128, 128, 3, 1),
        ("conv1x1_64_64_L1", 1, 64, 64, 1, 1), ("conv1x1_32_32_L0", 0, 32, 32, 1, 1), ("conv_32_1_L0", 0, 32, 1, 3, 1),
    ]
    for name, lv, Cin, Cout, ks, s in convs:
        if not (want(name + "_fwd") or want(name + "_wgrad")):
            continue
        D, Hh, Ww = L[lv]
        x = torch.randn(B, D, Hh, Ww, Cin, device=dev)
        w = torch.randn(Cout, Cin, ks, ks, ks, device=dev) * 0.05
        wp = pack(w, 0)
        Do, Ho, Wo = [(d + 2 * (ks // 2) - ks) // s + 1 for d in (D, Hh, Ww)]
        out = torch.empty(B, Do, Ho, Wo, Cout, device=dev)
        st = torch.empty(lib.raw("stx_conv3d_fwd_stat_rows")(B, D, Hh, Ww, Cin, Cout, ks, s), 2, Cout, device=dev)
        fl = 2.0 * B * Do * Ho * Wo * Cout * Cin * ks ** 3
        if want(name + "_fwd"):
            ms = timeit(lambda: lib.call("stx_conv3d_fwd", P(x), P(wp), P(out), None, None, None, P(st), B, D, Hh, Ww,
                                         Cin, Cout, ks, s, 0, stream()), it)
            report(name + "_fwd", ms, flops=fl)
        if Cout % 32 == 0 and Cin % 32 == 0 and not a.skip_wgrad and want(name + "_wgrad"):
            n = lib.raw("stx_conv3d_wgrad_workspace_floats")(B, Do, Ho, Wo, Cin, Cout, ks, s)
            ws = torch.empty(n, device=dev)
            dw = torch.empty(Cout, Cin, ks ** 3, device=dev)
            ms = timeit(lambda: lib.call("stx_conv3d_wgrad", P(x), P(out), P(dw), P(ws), B, D, Hh, Ww, Cin, Do, Ho, Wo,
                                         Cout, ks, s, stream()), it)
            report(name + "_wgrad", ms, flops=fl)
        del x, out

    if want("c1"):
        D, Hh, Ww = L[0]
        x = torch.randn(B, D, Hh, Ww, 32, device=dev)
        w = torch.randn(1, 32, 27, device=dev) * 0.05
        out = torch.empty(B, D, Hh, Ww, device=dev)
        ms = timeit(lambda: lib.call("stx_conv3d_c1_fwd", P(x), P(w), None, P(out), B, D, Hh, Ww, 32, stream()), it)
        report("conv_c1_32_1_L0_fwd", ms, nbytes=(x.numel() + out.numel()) * 4)
        ws = torch.empty(lib.raw("stx_conv3d_c1_wgrad_workspace_floats")(32), device=dev)
        dw = torch.empty(1, 32, 27, device=dev)
        ms = timeit(lambda: lib.call("stx_conv3d_c1_wgrad", P(x), P(out), P(dw), P(ws), B, D, Hh, Ww, 32, stream()), it)
        report("conv_c1_32_1_L0_wgrad", ms, nbytes=(x.numel() + out.numel()) * 4)
        ms = timeit(lambda: lib.call("stx_conv3d_c1_dgrad", P(out), P(w), P(x), B, D, Hh, Ww, 32, stream()), it)
        report("conv_c1_32_1_L0_dgrad", ms, nbytes=(x.numel() + out.numel()) * 4)
        del x, out

    for name, lv, Cin, Cout in [("deconv_128_64_L2", 2, 128, 64), ("deconv_64_32_L1", 1, 64, 32)]:
        if not want(name + "_fwd"):
            continue
        D, Hh, Ww = L[lv]
        x = torch.randn(B, D, Hh, Ww, Cin, device=dev)
        w = torch.randn(Cin, Cout, 3, 3, 3, device=dev) * 0.05
        wp = pack(w, 2)
        out = torch.empty(B, 2 * D, 2 * Hh, 2 * Ww, Cout, device=dev)
        fl = 2.0 * B * D * Hh * Ww * Cout * Cin * 27
        ms = timeit(lambda: lib.call("stx_deconv3d_fwd", P(x), P(wp), P(out), None, None, None, None, B, D, Hh, Ww, Cin,
                                     Cout, 2 * D, 2 * Hh, 2 * Ww, 0, stream()), it)
        report(name + "_fwd", ms, flops=fl)

    if want("bn"):
        D, Hh, Ww = L[0]
        nvox, C = B * D * Hh * Ww, 32
        z = torch.randn(nvox, C, device=dev)
        y = torch.empty_like(z)
        sc, sh = torch.rand(C, device=dev), torch.rand(C, device=dev)
        ms = timeit(lambda: lib.call("stx_bn_apply", P(z), P(sc), P(sh), None, None, None, P(y), nvox, C, 1, 1, stream()), it)
        report("bn_apply_L0", ms, nbytes=z.numel() * 8)
        nrows = lib.raw("stx_conv3d_fwd_stat_rows")(B, D, Hh, Ww, C, C, 3, 1)
        fpart = torch.randn(nrows, 2, C, device=dev)
        fo = [torch.empty(C, device=dev) for _ in range(6)]
        ms = timeit(lambda: lib.call("stx_bn_finalize", P(fpart), nrows, C, float(nvox), P(sc), P(sh), P(fo[4]), P(fo[5]),
                                     0.1, 1e-5, P(fo[0]), P(fo[1]), P(fo[2]), P(fo[3]), stream()), it)
        report("bn_finalize_L0", ms, nbytes=fpart.numel() * 4)
        NB = lib.raw("stx_bn_reduce_blocks")()
        part, sums = torch.empty(NB, 3, C, device=dev), torch.empty(3, C, device=dev)
        ms = timeit(lambda: lib.call("stx_bn_bwd_reduce", P(z), P(y), P(z), P(sc), P(sh), None, None, None, P(part),
                                     P(sums), nvox, C, 1, stream()), it)
        report("bn_bwd_reduce_L0", ms, nbytes=z.numel() * 12)
        dz = torch.empty_like(z)
        ms = timeit(lambda: lib.call("stx_bn_bwd_apply", P(z), P(y), P(z), P(sc), P(sh), P(sc), None, None, None, None,
                                     P(sums), P(dz), None, None, nvox, C, 1, stream()), it)
        report("bn_bwd_apply_L0", ms, nbytes=z.numel() * 16)

    if want("head"):
        cost = torch.randn(B, D4, H4, W4, device=dev) * 3
        disp, stats = torch.empty(B, a.H, a.W, device=dev), torch.empty(B, a.H, a.W, 2, device=dev)
        ms = timeit(lambda: lib.call("stx_head_fwd", P(cost), P(disp), P(stats), B, D4, H4, W4, a.D, a.H, a.W, stream()), it)
        report("head_fwd", ms, nbytes=(cost.numel() + disp.numel()) * 4)
        g, gc = torch.randn_like(disp), torch.empty_like(cost)
        ws = torch.empty(lib.raw("stx_head_bwd_workspace_floats")(B, D4, a.H, a.W), device=dev)
        ms = timeit(lambda: lib.call("stx_head_bwd", P(g), P(cost), P(disp), P(stats), P(gc), P(ws), B, D4, H4, W4, a.D,
                                     a.H, a.W, stream()), it)
        report("head_bwd", ms, nbytes=(cost.numel() * 2 + disp.numel() * 4) * 4)

    if want("dwconv"):
        # ACVNet patch convolutions (acv.py:183-187): 40-channel attention volume, dilations 1 / 2 / 3 per channel slice
        D0, H0, W0 = L[0]
        C = 40
        x = torch.randn(B, D0, H0, W0, C, device=dev)
        y = torch.empty_like(x)
        w = torch.randn(C, 9, device=dev)
        dil = torch.tensor([1, 1, 2, 2, 2, 2, 3, 3, 3, 3], dtype=torch.int32, device=dev)
        ms = timeit(lambda: lib.call("stx_dwconv_hw_fwd", P(x), P(w), P(dil), P(y), B, D0, H0, W0, C, 0, stream()), it)
        report("dwconv_hw_acv_patch_fwd", ms, nbytes=x.numel() * 8)
        ws = torch.empty(lib.raw("stx_dwconv_hw_wgrad_workspace_floats")(C), device=dev)
        gw = torch.empty(C, 9, device=dev)
        ms = timeit(lambda: lib.call("stx_dwconv_hw_wgrad", P(x), P(y), P(dil), P(gw), P(ws), B, D0, H0, W0, C, stream()), it)
        report("dwconv_hw_acv_patch_wgrad", ms, nbytes=x.numel() * 8)

    if want("mish"):
        D0, H0, W0 = L[0]
        x = torch.randn(B, D0, H0, W0, 32, device=dev)
        y, g = torch.empty_like(x), torch.randn(B, D0, H0, W0, 32, device=dev)
        ms = timeit(lambda: lib.call("stx_mish_fwd", P(x), P(y), x.numel(), stream()), it)
        report("mish_fwd_L0", ms, nbytes=x.numel() * 8)
        ms = timeit(lambda: lib.call("stx_mish_bwd", P(g), P(x), P(y), x.numel(), stream()), it)
        report("mish_bwd_L0", ms, nbytes=x.numel() * 12)
        del x, y, g

    if want("estimator"):
        # full-resolution probability volume [B,192,H,W] (425 MB): two Gaussian modes per pixel + rough floor
        d = torch.arange(a.D, device=dev, dtype=torch.float32).view(1, a.D, 1, 1)
        m1, m2 = torch.rand(B, 1, a.H, a.W, device=dev) * (a.D - 1), torch.rand(B, 1, a.H, a.W, device=dev) * (a.D - 1)
        x = 0.6 * torch.exp(-0.5 * ((d - m1) / 1.5) ** 2) + 0.4 * torch.exp(-0.5 * ((d - m2) / 2.0) ** 2)
        x += torch.rand(B, a.D, a.H, a.W, device=dev) * 1e-3
        x /= x.sum(1, keepdim=True)
        out = torch.empty(B, a.H, a.W, device=dev)
        for entry in ("stx_softargmax_fwd", "stx_unimodal_fwd", "stx_dominant_modal_fwd"):
            ms = timeit(lambda: lib.call(entry, P(x), P(out), B, a.D, a.H * a.W, stream()), it)
            report("estimator_" + entry[4:-4], ms, nbytes=(x.numel() + out.numel()) * 4)
        del x


if __name__ == "__main__":
    main()
