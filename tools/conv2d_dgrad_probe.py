"""Probe (GPU): the data gradient of the 2-D feature CNN's 3x3 stride-1 convolutions two ways on MIOpen --
  (a) aten.convolution_backward(input only): MIOpen's backward-data solvers (in the step trace: igemm_bwd_gtcx35 ... gkgs,
      an atomic split-K kernel behind a zero fill of its output);
  (b) the FORWARD solver on the flipped, channel-transposed weight: dgrad of a stride-1 convolution IS a convolution.
Also times forward and the weight gradient for the table.  Channels-last fp32, both views batched (B = 2), cudnn.benchmark on,
the shipped find-db.  One JSON line per layer shape.  Measurement tool: not part of the product path."""
import json
import sys

import torch
import torch.nn.functional as F


def timed(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    from stereo_toolbox_amd.utils import use_tuning_db
    use_tuning_db()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    shapes = [("layer1 32->32 288x480", 2, 32, 32, 288, 480, 1), ("layer2 64->64 144x240", 2, 64, 64, 144, 240, 1),
              ("layer3 128->128 144x240", 2, 128, 128, 144, 240, 1), ("layer4 128->128 144x240 dil2", 2, 128, 128, 144, 240, 2),
              ("layer3.0 64->128 144x240", 2, 64, 128, 144, 240, 1)]
    if len(sys.argv) > 1 and sys.argv[1] == "b1":
        shapes = [(n, 1, a, b, h, w, d) for n, _, a, b, h, w, d in shapes]
    for name, B, Ci, Co, H, W, dil in shapes:
        x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, Co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        pad = (dil, dil)

        def fwd():
            return F.conv2d(x, w, None, 1, pad, dil)

        def dgrad_bwd():
            return torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), pad, (dil, dil), False, (0, 0), 1,
                                                       (True, False, False))[0]

        def flipw():
            return w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)

        wf = flipw()

        def dgrad_fwd():
            return F.conv2d(gy, wf, None, 1, pad, dil)

        def wgrad():
            return torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), pad, (dil, dil), False, (0, 0), 1,
                                                       (False, True, False))[1]

        a, b = dgrad_bwd(), dgrad_fwd()
        err = (a - b).abs().max().item() / a.abs().max().item()
        flop = 2.0 * B * H * W * Ci * Co * 9
        rec = {"layer": name, "B": B, "fwd_us": timed(fwd), "dgrad_bwd_solver_us": timed(dgrad_bwd),
               "dgrad_fwd_solver_us": timed(dgrad_fwd), "flip_weight_us": timed(flipw), "wgrad_us": timed(wgrad),
               "dgrad_rel_diff": err, "gflop": flop / 1e9}
        for k in ("fwd_us", "dgrad_bwd_solver_us", "dgrad_fwd_solver_us", "wgrad_us"):
            rec[k.replace("_us", "_frac_of_157T")] = round(flop / (rec[k] * 1e-6) / 157.3e12, 3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
