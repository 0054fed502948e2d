"""`torch.ops.stx.*`: the hot path's kernels as dispatcher-visible PyTorch operators.

    import stereo_toolbox_amd.torch_ext as tx
    ops = tx.load()                                  # torch.utils.cpp_extension.load(...) -> torch.ops.stx
    vol = ops.cost_volume(Lg, Rg, Lc, Rc, 48, 40, True)          # differentiable

north_star / SURVEY.md 8(b) name a `TORCH_LIBRARY` module built with `torch.utils.cpp_extension` as the loader of the
kernels.  The product's default binding is ctypes over the torch-free C-ABI (`_capi.py`); this module
is the other loader on the SAME shared object: `csrc/torch_binding.cpp` registers schemas, ROCm-device kernels, Meta (shape)
kernels -- so FakeTensor / `torch.compile` tracing and `device="meta"` work -- and autograd for the volume builder, and is
compiled in-tree (`stereo_toolbox_amd/lib/torch_ext/`, git-ignored like the library) against `include/stx_hip.h`, linking
`lib/libstx_hip.so`.  No kernel and no fallback lives here: CPU tensors get the dispatcher's "could not run ... 'CPU'
backend" error.
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_OPS = None


def build(verbose=False):
    """Compile (or re-use) the extension module in-tree and import it; returns the python module object."""
    from torch.utils.cpp_extension import load as _load

    from .build import LIBDIR, build_hip
    build_hip(verbose=False)
    bdir = os.path.join(LIBDIR, "torch_ext")
    os.makedirs(bdir, exist_ok=True)
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    return _load(name="stx_torch_ext", sources=[os.path.join(_HERE, "csrc", "torch_binding.cpp")], build_directory=bdir,
                 extra_ldflags=[f"-L{LIBDIR}", "-l:libstx_hip.so", f"-Wl,-rpath,{LIBDIR}"], with_cuda=True, verbose=verbose)


def load(verbose=False):
    """Build / load once per process and return `torch.ops.stx`."""
    global _OPS
    if _OPS is None:
        build(verbose)
        _OPS = torch.ops.stx
    return _OPS
