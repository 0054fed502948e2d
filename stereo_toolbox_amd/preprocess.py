"""Device-side input step of the evaluation path.

Reference: every dataset's test branch calls `pad_to_2x` (datasets/data_augmentation/__init__.py:57-80) on the uint8
images and then `get_transform()` (datasets/utils.py:62-69: torchvision ToTensor + Normalize with the ImageNet
statistics) on the CPU, per image, inside the DataLoader workers.  Here the uint8 HWC image goes to the device as it
is and ONE hand-written kernel (csrc/preprocess.hip) emits the padded, normalised NCHW float tensor; ground truth and
masks are padded with a stock `F.pad` (zeros, like the reference).

  prepare_pair(left_u8, right_u8, disp=None, mask=None) -> left, right [B,3,Hp,Wp] float32 (+ padded disp / mask)
  pad_to_2x(left, right, disp=None, mask=None)           -> the reference function on device tensors (no normalisation)
  unpad(pred, H, W)                                      -> crop a prediction back to the original image
"""
import ctypes
import math

import torch
import torch.nn.functional as F

from . import ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # datasets/utils.py:63
IMAGENET_STD = (0.229, 0.224, 0.225)       # datasets/utils.py:64
PAD_SCALE = 96                             # datasets/data_augmentation/__init__.py:63


def padded_size(H, W, scale=PAD_SCALE):
    """(Hp, Wp, top_pad, right_pad) of the reference's pad_to_2x."""
    Hp, Wp = int(math.ceil(H / scale) * scale), int(math.ceil(W / scale) * scale)
    return Hp, Wp, Hp - H, Wp - W


def _as_batch(img):
    if img.dim() == 3:
        img = img.unsqueeze(0)
    if img.dim() != 4 or img.shape[-1] != 3:
        raise ops.StxError(f"expected a uint8 image [H,W,3] or [B,H,W,3], got {tuple(img.shape)}")
    return img.contiguous()


def pad_normalize(img_u8, scale=PAD_SCALE, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """uint8 [H,W,3] / [B,H,W,3] on the device -> float32 [B,3,Hp,Wp]: pad_to_2x + ToTensor + Normalize, one kernel."""
    img = _as_batch(img_u8)
    if not img.is_cuda or img.dtype != torch.uint8:
        raise ops.StxError(f"pad_normalize: expected a uint8 ROCm device tensor, got {img.dtype} on {img.device} "
                           "(no CPU fallback; the CPU restatement lives in oracle/ and is test-only)")
    B, H, W, _ = img.shape
    Hp, Wp, top, _ = padded_size(H, W, scale)
    out = torch.empty(B, 3, Hp, Wp, dtype=torch.float32, device=img.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    ops._call("stx_pad_normalize_u8", ops._p(img), ops._p(out), B, H, W, Hp, Wp, top, m, s)
    return out


def _pad_map(t, top, right):
    """disp [.., H, W] / distribution [.., D, H, W] / mask: zeros on top and to the right (reference :70-78)."""
    return None if t is None else F.pad(t, (0, right, top, 0))


def pad_to_2x(left, right, disp=None, mask=None, scale=PAD_SCALE):
    """The reference function on device tensors: left/right [H,W,C] or [B,H,W,C] (any dtype) are zero-padded on top and to
    the right to multiples of `scale`; disp (2-D disparity or 3-D distribution) and mask likewise.  No normalisation."""
    H, W = left.shape[-3], left.shape[-2]
    _, _, top, rp = padded_size(H, W, scale)
    if mask is not None:
        assert disp is not None and disp.dim() == 2        # reference :76-77
    pad_img = lambda t: F.pad(t, (0, 0, 0, rp, top, 0))
    return pad_img(left), pad_img(right), _pad_map(disp, top, rp), _pad_map(mask, top, rp)


def prepare_pair(left_u8, right_u8, disp=None, mask=None, scale=PAD_SCALE):
    """Everything the reference's test-time `__getitem__` does to a pair (kitti.py:84-101 etc.) on the device:
    -> dict(left, right [B,3,Hp,Wp] float32 normalised, gt_disp, noc_mask padded float32 or None, pad=(top, right))."""
    H, W = left_u8.shape[-3], left_u8.shape[-2]
    _, _, top, rp = padded_size(H, W, scale)
    return {"left": pad_normalize(left_u8, scale), "right": pad_normalize(right_u8, scale),
            "gt_disp": None if disp is None else _pad_map(disp.float(), top, rp),
            "noc_mask": None if mask is None else _pad_map(mask.float(), top, rp), "pad": (top, rp)}


def unpad(pred, H, W):
    """Crop a prediction [.., Hp, Wp] made on a padded pair back to the original H x W (bottom-left aligned)."""
    Hp = pred.shape[-2]
    return pred[..., Hp - H:, :W]
