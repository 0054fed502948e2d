"""Cases and seeded inputs of the f-1 fixtures (tests/golden/f1ops.npz): shared by the generator (make_golden_f1ops.py, which
imports the reference and runs only in the build container) and by tests/test_f1ops.py (which must not: /root/reference does not
exist on the GPU box)."""
import torch

from stereo_toolbox_amd.utils import synthetic_tensor

# (name, B, C, H, W, disparity range)
WARP_CASES = (("a", 2, 5, 9, 23, -3.0, 12.0), ("b", 1, 3, 6, 70, 0.0, 30.0))
# (name, B, C, H, W, maxdisp, groups)
CORR_CASES = (("a", 2, 8, 5, 37, 6, 2), ("b", 1, 4, 3, 52, 24, 1), ("c", 1, 6, 2, 140, 9, 3))
# (name, B, D, H, W)
VAR_CASES = (("a", 2, 12, 5, 7), ("b", 1, 24, 3, 9))


def warp_inputs(B, C, H, W, lo, hi, seed):
    return (synthetic_tensor((B, C, H, W), seed), synthetic_tensor((B, 1, H, W), seed + 1, lo=lo, hi=hi),
            synthetic_tensor((B, C, H, W), seed + 2))


def corr_inputs(B, C, H, W, md, G, seed):
    return (synthetic_tensor((B, C, H, W), seed), synthetic_tensor((B, C, H, W), seed + 1),
            synthetic_tensor((B, G, 2 * md + 1, H, W), seed + 2))


def var_inputs(B, D, H, W, seed):
    x = torch.softmax(2.0 * synthetic_tensor((B, D, H, W), seed), dim=1)
    disp = synthetic_tensor((B, 1, H, W), seed + 1, lo=0.0, hi=float(D - 1))
    samples = synthetic_tensor((B, D, H, W), seed + 2, lo=0.0, hi=float(4 * D))
    return x, disp, samples, synthetic_tensor((B, 1, H, W), seed + 3)
