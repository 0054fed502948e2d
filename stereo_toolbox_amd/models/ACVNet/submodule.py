"""Drop-in API of reference models/ACVNet/submodule.py for the pieces ACVNet uses
(convbn_3d, disparity_regression, build_gwc_volume, build_concat_volume [ACV semantics],
groupwise_correlation, attention_block), hot-path pieces backed by the HIP kernels."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..features2d import BasicBlock, convbn  # noqa: F401
from ..GwcNet.submodule import build_gwc_volume, convbn_3d, disparity_regression, groupwise_correlation  # noqa: F401


def build_concat_volume(refimg_fea, targetimg_fea, maxdisp):
    """reference ACVNet/submodule.py:180-191: left feature copied to every column (NOT masked where
    w < d), right feature shifted and masked -> [B, 2C, D, H, W]."""
    return ops.to_ncdhw(ops.cost_volume(None, None, refimg_fea, targetimg_fea, maxdisp, 0, mask_left=False))


class attention_block(nn.Module):
    """Windowed 3-D self attention (reference ACVNet/submodule.py:383-429), < 1 % of the FLOPs: stays on
    stock PyTorch-ROCm ops (rocBLAS GEMMs + softmax), per SURVEY.md 8a row a10.  Takes and returns
    channels-last [B, D, H, W, C] activations; parameter names/shapes equal the reference's
    (`qkv_3d` Linear, `final1x1` Conv3d 1x1x1 with bias)."""

    def __init__(self, channels_3d, num_heads=8, block=4):
        super().__init__()
        self.block = block
        self.dim_3d = channels_3d
        self.num_heads = num_heads
        head_dim_3d = self.dim_3d // num_heads
        self.scale_3d = head_dim_3d ** -0.5
        self.qkv_3d = nn.Linear(self.dim_3d, self.dim_3d * 3, bias=True)
        self.final1x1 = torch.nn.Conv3d(self.dim_3d, self.dim_3d, 1)

    def _pad_bias(self, H, W, pad_b, pad_r, d, device):
        """Additive attention bias [1, d*h*w, nb, nb] that separates padded from real voxels inside a window:
        -1000 between two tokens of different kind, 0 otherwise (reference submodule.py:405-413).  A token is
        "padded" if it lies in the bottom pad rows or the right pad columns; the reference's slices `-pad_b:` /
        `-pad_r:` select EVERYTHING when that pad is 0, i.e. with padding on one axis only every token counts as
        padded and the bias vanishes -- kept, it is what trained checkpoints saw.  Cached per shape."""
        key = (H, W, pad_b, pad_r, d, str(device))
        hit = self.__dict__.get("_stx_bias")
        if hit is not None and hit[0] == key:
            return hit[1]
        b0, b1, b2 = self.block
        rows = torch.arange(H, device=device) >= (H - pad_b if pad_b > 0 else 0)
        cols = torch.arange(W, device=device) >= (W - pad_r if pad_r > 0 else 0)
        padded = rows.view(H, 1) | cols.view(1, W)                                           # [H, W] bool
        win = padded.view(H // b1, b1, W // b2, b2).permute(0, 2, 1, 3).reshape(-1, b1 * b2)   # [h*w, b1*b2]
        differ = win.unsqueeze(2) != win.unsqueeze(1)                                        # [h*w, b1*b2, b1*b2]
        bias = torch.where(differ, -1000.0, 0.0).to(torch.float32)
        bias = bias.repeat(d, b0, b0).unsqueeze(0)             # every window along D, every (dz, dz') pair of a window
        self.__dict__["_stx_bias"] = (key, bias)
        return bias

    def forward(self, x):
        B, D, H0, W0, C = x.shape
        b0, b1, b2 = self.block
        pad_r = (b2 - W0 % b2) % b2
        pad_b = (b1 - H0 % b1) % b1
        if pad_r or pad_b:
            x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        _, _, H, W, _ = x.shape
        d, h, w = D // b0, H // b1, W // b2
        nb = b0 * b1 * b2
        hd = C // self.num_heads
        # windows: [B, d, h, w, nb, C]
        xw = x.view(B, d, b0, h, b1, w, b2, C).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(B, d * h * w, nb, C)
        qkv = self.qkv_3d(xw).reshape(B, d * h * w, nb, 3, self.num_heads, hd).permute(3, 0, 1, 4, 2, 5)
        q, k, v = qkv[0], qkv[1], qkv[2]                                  # [B, windows, heads, nb, hd]
        attn = (q @ k.transpose(-2, -1)) * self.scale_3d
        if pad_r > 0 or pad_b > 0:
            attn = attn + self._pad_bias(H, W, pad_b, pad_r, d, x.device).unsqueeze(2)
        attn = torch.softmax(attn, dim=-1)
        y = attn @ v                                                      # [B, windows, heads, nb, hd]
        # channel index of the reference after its permute/reshape is (head, hd)
        y = y.view(B, d, h, w, self.num_heads, b0, b1, b2, hd).permute(0, 1, 5, 2, 6, 3, 7, 4, 8)
        y = y.reshape(B, D, H, W, C)
        if pad_r > 0 or pad_b > 0:
            y = y[:, :, :H0, :W0, :]
        return F.linear(y, self.final1x1.weight.view(C, C), self.final1x1.bias).contiguous()
