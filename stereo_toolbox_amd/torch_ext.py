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

Never waits on somebody else's build (round 5 lost its GPU suite to this: `cpp_extension.load` spins in
`FileBaton.wait()` for as long as a `lock` file exists in its build directory, and an interrupted build had left one in
the tree that travelled to the GPU box):

* a finished module `lib/torch_ext/stx_torch_ext.so` whose stamp (`stx_torch_ext.stamp`: sha256 of the binding's source,
  of include/stx_hip.h and of the torch version) matches is loaded with `torch.ops.load_library` -- no ninja, no baton;
* otherwise `cpp_extension.load` builds into a directory only THIS process uses (`lib/obj/torch_ext.<pid>.<n>`, under the
  gpurun-ignored `lib/obj/`), so the only baton there is our own; the module is then moved into `lib/torch_ext/` with an
  atomic rename and the scratch directory removed (also from a `finally`, and stale ones of dead processes are swept).
"""
import hashlib
import os
import shutil

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "torch_binding.cpp")
_HDR = os.path.join(os.path.dirname(_HERE), "include", "stx_hip.h")
_NAME = "stx_torch_ext"
_OPS = None
_LOADED = False


def _paths():
    from .build import LIBDIR
    d = os.path.join(LIBDIR, "torch_ext")
    return d, os.path.join(d, _NAME + ".so"), os.path.join(d, _NAME + ".stamp")


def _stamp():
    h = hashlib.sha256()
    for p in (_SRC, _HDR):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(torch.__version__.encode())
    return h.hexdigest()


def is_current():
    """True when lib/torch_ext/ holds a finished module built from the present sources (no build needed)."""
    _, so, stamp = _paths()
    try:
        with open(stamp) as f:
            return os.path.exists(so) and f.read().strip() == _stamp()
    except OSError:
        return False


def _sweep(objdir):
    """Scratch build directories whose owner process is gone (an interrupted build): remove them."""
    try:
        names = os.listdir(objdir)
    except OSError:
        return
    for n in names:
        if not n.startswith("torch_ext."):
            continue
        try:
            pid = int(n.split(".")[1])
            os.kill(pid, 0)                                  # raises if no such process
            alive = pid != os.getpid()
        except (ValueError, IndexError, ProcessLookupError):
            alive = False
        except PermissionError:
            alive = True
        if not alive:
            shutil.rmtree(os.path.join(objdir, n), ignore_errors=True)


def build(verbose=False, force=False):
    """Make lib/torch_ext/stx_torch_ext.so current (compile with torch.utils.cpp_extension if it is not); returns its path.
    A compile also dlopens the fresh module in this process (that is what `cpp_extension.load` does): `_LOADED` records it."""
    global _LOADED
    from .build import LIBDIR, build_hip
    build_hip(verbose=False)
    d, so, stamp = _paths()
    if is_current() and not force:
        return so
    from torch.utils.cpp_extension import load as _load
    os.makedirs(d, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    _sweep(objdir)
    n = 0
    while os.path.exists(scratch := os.path.join(objdir, f"torch_ext.{os.getpid()}.{n}")):
        n += 1
    os.makedirs(scratch)
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    try:
        # is_python_module=False: build, then torch.ops.load_library(<scratch>/stx_torch_ext.so) -- the TORCH_LIBRARY
        # registrations of this process come from that mapping (it outlives the directory)
        _load(name=_NAME, sources=[_SRC], build_directory=scratch, is_python_module=False,
              extra_ldflags=[f"-L{LIBDIR}", "-l:libstx_hip.so", f"-Wl,-rpath,{LIBDIR}"], with_cuda=True, verbose=verbose)
        _LOADED = True
        tmp = so + f".{os.getpid()}.tmp"
        shutil.copyfile(os.path.join(scratch, _NAME + ".so"), tmp)
        os.replace(tmp, so)                                  # atomic: a concurrent reader sees the old or the new module
        with open(stamp + f".{os.getpid()}.tmp", "w") as f:
            f.write(_stamp() + "\n")
        os.replace(stamp + f".{os.getpid()}.tmp", stamp)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    return so


def load(verbose=False):
    """Build if needed, dlopen the module once per process, return `torch.ops.stx`."""
    global _OPS, _LOADED
    if _OPS is None:
        so = build(verbose)
        if not _LOADED:
            from ._capi import get_lib
            get_lib()                                        # libstx_hip.so mapped first: the module's DT_NEEDED entry matches its
            torch.ops.load_library(so)                       # SONAME wherever the tree lives (the rpath is this box's absolute path)
            _LOADED = True
        _OPS = torch.ops.stx
    return _OPS
