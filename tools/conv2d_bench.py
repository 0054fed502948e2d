"""GPU timing of csrc/conv2d.hip against MIOpen on the 2-D feature CNN's 3x3 stride-1 shapes (both views batched, channels-last,
fp32): forward and data gradient, back to back (HIP events around 20 launches) and behind a 512 MiB fill (cold L2 / MALL).
One JSON line per shape.  Measurement tool: not part of the product path."""
import ctypes
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from stereo_toolbox_amd._capi import get_lib          # noqa: E402
from stereo_toolbox_amd.utils import use_tuning_db    # noqa: E402


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def timed(fn, iters=20, warm=5, flush=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if flush is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    tot = 0.0
    for _ in range(iters):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


def main():
    use_tuning_db()
    torch.backends.cudnn.benchmark = True
    lib = get_lib()
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    flush = torch.empty(128 << 20, dtype=torch.float32, device=dev)
    shapes = [("layer2 64->64 144x240", 2, 64, 64, 144, 240), ("layer1 32->32 288x480", 2, 32, 32, 288, 480),
              ("layer3.0 64->128 144x240", 2, 64, 128, 144, 240), ("kitti layer2 64->64 96x312", 2, 64, 64, 96, 312),
              ("layer2 64->64 144x240 B=4", 4, 64, 64, 144, 240)]
    for name, B, Ci, Co, H, W in shapes:
        x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, Co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        xn, gyn = x.permute(0, 2, 3, 1), gy.permute(0, 2, 3, 1)
        assert xn.is_contiguous() and gyn.is_contiguous()
        wo = w.permute(0, 2, 3, 1)
        assert wo.is_contiguous()
        out = torch.empty(B, H, W, Co, device=dev)
        rows = lib.raw("stx_conv2d_stat_rows")(2)
        stats = torch.empty(rows, 2, Co, device=dev)

        def mine():
            lib.call("stx_conv2d_fwd", P(xn), P(wo), P(out), None, B, H, W, Ci, Co, 0, 1, st)

        def mine_stats():
            lib.call("stx_conv2d_fwd", P(xn), P(wo), P(out), P(stats), B, H, W, Ci, Co, 0, 2, st)

        def ref():
            return F.conv2d(x, w, None, 1, 1)

        mine()
        r = ref()
        err = (out.permute(0, 3, 1, 2) - r).abs().max().item() / r.abs().max().item()
        flop = 2.0 * B * H * W * Ci * Co * 9
        rec = {"shape": name, "gflop": flop / 1e9, "fwd_rel_err": err,
               "fwd_hip_us": timed(mine), "fwd_hip_stats_us": timed(mine_stats), "fwd_miopen_us": timed(ref),
               "fwd_hip_cold_us": timed(mine, flush=flush), "fwd_miopen_cold_us": timed(ref, flush=flush)}
        if Co in (32, 64):
            gx = torch.empty(B, H, W, Ci, device=dev)

            def mine_d():
                lib.call("stx_conv2d_fwd", P(gyn), P(wo), P(gx), None, B, H, W, Co, Ci, 1, 1, st)

            def ref_d():
                return torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                           (True, False, False))[0]
            mine_d()
            rd = ref_d()
            rec["dgrad_rel_err"] = (gx.permute(0, 3, 1, 2) - rd).abs().max().item() / rd.abs().max().item()
            rec.update({"dgrad_hip_us": timed(mine_d), "dgrad_miopen_us": timed(ref_d),
                        "dgrad_hip_cold_us": timed(mine_d, flush=flush), "dgrad_miopen_cold_us": timed(ref_d, flush=flush)})
        for k in [k for k in rec if k.endswith("_us")]:
            rec[k.replace("_us", "_frac")] = round(flop / (rec[k] * 1e-6) / 157.3e12, 3)
            rec[k] = round(rec[k], 2)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
