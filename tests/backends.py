"""Test backends: the same kernel-level parity tests run
  * on the host SIMT emulator build of the kernel sources (CPU tensors, `-m "not gpu"`), and
  * on the gfx950 library through the C-ABI with ROCm tensors (`-m gpu`).
Both are compared against the CPU oracle (oracle/torch_oracle.py or plain torch fp32 ops).
"""
import ctypes

import pytest
import torch


class Backend:
    def __init__(self, name):
        self.name = name
        if name == "emu":
            from tests.emu_util import emu_lib
            self.lib = emu_lib()
            self.device = torch.device("cpu")
        else:
            from stereo_toolbox_amd._capi import get_lib
            if not torch.cuda.is_available():
                pytest.skip("no ROCm device")
            self.lib = get_lib()
            self.device = torch.device("cuda:0")

    @property
    def stream(self):
        if self.name == "emu":
            return None
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def dev(self, t):
        return None if t is None else t.to(self.device).contiguous()

    def empty(self, *shape, dtype=torch.float32, fill=float("nan")):
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        if dtype.is_floating_point:
            t.fill_(fill)
        else:
            t.zero_()
        return t

    def call(self, name, *args):
        """Tensors may be passed directly; they stay referenced for the duration of the call."""
        conv = [ctypes.c_void_p(a.data_ptr()) if isinstance(a, torch.Tensor) else a for a in args]
        self.lib.call(name, *conv, self.stream)

    def raw(self, name):
        return self.lib.raw(name)


def ptr(t):
    """Identity marker for pointer arguments: Backend.call converts tensors to device pointers."""
    return t


def ndhwc(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(t):
    return t.permute(0, 4, 1, 2, 3).contiguous()


BACKENDS = ["emu", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return Backend(request.param)


@pytest.fixture
def tune(be):
    """Set a tuning / A-B switch of the library under test for the duration of a test (stx_set_tuning; the library reads
    its STX_* environment variables only once, when it is loaded)."""
    saved = {}

    def set_(name, value):
        old = be.lib.set_tuning(name, value)
        saved.setdefault(name, old)
    yield set_
    for name, old in saved.items():
        be.lib.set_tuning(name, old)
