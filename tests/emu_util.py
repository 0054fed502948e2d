"""Helpers to drive the host-emulator build of the HIP kernels from CPU tensors (tests only)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))

_EMU = None


def emu_lib():
    global _EMU
    if _EMU is None:
        from build_emu import build_emu
        from stereo_toolbox_amd._capi import StxLib
        _EMU = StxLib(build_emu())
    return _EMU


def ptr(t):
    if t is None:
        return None
    assert t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last_3d), "non-dense tensor"
    return ctypes.c_void_p(t.data_ptr())


def ndhwc(t):
    """[B,C,D,H,W] -> dense [B,D,H,W,C] copy."""
    return t.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(t):
    """dense [B,D,H,W,C] -> [B,C,D,H,W] contiguous copy."""
    return t.permute(0, 4, 1, 2, 3).contiguous()


class emu_product_path:
    """Context manager (tests only): run the *product* host code (ops.py / aggregation.py / models)
    against the host-emulator build of the kernels with CPU tensors, by monkeypatching the four
    device-specific hooks of stereo_toolbox_amd.ops.  Lets the whole autograd wiring be checked
    against the oracle without a GPU.  Nothing in the product references this."""

    def __enter__(self):
        from stereo_toolbox_amd import ops
        self.ops = ops
        self.saved = (ops.get_lib, ops._chk, ops._stream, ops.on_device)
        ops.on_device = lambda t: True
        lib = emu_lib()
        ops.get_lib = lambda: lib

        def chk(t, name, dims=None):
            if t is None:
                return
            assert t.dtype == torch.float32 and t.is_contiguous(), name
            assert dims is None or t.dim() == dims, name
        ops._chk = chk
        ops._stream = lambda dev=None: None
        return self

    def __exit__(self, *exc):
        self.ops.get_lib, self.ops._chk, self.ops._stream, self.ops.on_device = self.saved
        return False
