"""Shapes and inputs shared by tests/golden/make_golden_foundation.py (imports the reference) and tests/test_foundation.py."""
import torch

from stereo_toolbox_amd.utils import synthetic_tensor

# tag -> (B, C, G, H, W, D', Cc): gwc feature channels / groups, 1/4-resolution size, disparities, concat channels per side
CASES = {
    "vitl_224ch": (1, 224, 8, 4, 40, 12, 12),
    "vitb_160ch": (1, 160, 8, 3, 36, 8, 12),
    "vits_128ch": (2, 128, 8, 4, 24, 6, 12),
    "generic_32ch": (2, 32, 4, 5, 21, 7, 4),
}


def inputs(tag):
    """(L, R, Lc, Rc, gw, gc): feature maps, concat maps, the weights of the scalar whose gradient is stored."""
    B, C, G, H, W, D, Cc = CASES[tag]
    s = sum(ord(c) for c in tag)
    L, R = synthetic_tensor((B, C, H, W), s + 1), synthetic_tensor((B, C, H, W), s + 2)
    if tag == "generic_32ch":               # an all-zero group at a few pixels: the eps branch of F.normalize (norm clamped to 1e-12)
        L = L.clone()
        L[:, :8, 1, 3:6] = 0.0
        R = R.clone()
        R[:, 8:16, 2, 0:2] = 0.0
    Lc, Rc = synthetic_tensor((B, Cc, H, W), s + 3), synthetic_tensor((B, Cc, H, W), s + 4)
    gw = synthetic_tensor((B, G, D, H, W), s + 5)
    gc = synthetic_tensor((B, 2 * Cc, D, H, W), s + 6)
    return L, R, Lc, Rc, gw, gc
