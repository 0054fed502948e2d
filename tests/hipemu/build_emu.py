"""Compiles stereo_toolbox_amd/csrc/*.hip for the HOST with the SIMT emulator (tests only).

Output: tests/hipemu/_build/libstx_emu.so with the same C-ABI as the gfx950 library, so the
CPU test-suite can drive the very same kernel sources through ctypes on numpy/torch-CPU buffers.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "stereo_toolbox_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libstx_emu.so")
CXX = os.environ.get("STX_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-pthread", "-DSTX_HIPEMU", "-ffp-contract=off",
         "-Wno-unused-value", "-Wno-deprecated-declarations", "-I", HERE, "-I", CSRC]


def asan_runtime():
    """Shared AddressSanitizer runtime of the emulator's compiler (to LD_PRELOAD into the python process)."""
    r = subprocess.run([CXX, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    p = os.path.realpath(r.stdout.strip())
    return p if os.path.isfile(p) else None


def build_emu(force=False, verbose=False, asan=None):
    """asan (default: environment STX_EMU_ASAN=1): AddressSanitizer build into _build_asan/ -- out-of-bounds reads and
    writes of kernel code on the tensors' host buffers and on the emulated LDS abort the process (SURVEY.md section 5:
    the GPU would read / write past the buffer silently).  The python process must LD_PRELOAD asan_runtime()."""
    global OUT, LIB
    if asan is None:
        asan = os.environ.get("STX_EMU_ASAN") == "1"
    flags = list(FLAGS)
    if asan:
        OUT = os.path.join(HERE, "_build_asan")
        LIB = os.path.join(OUT, "libstx_emu.so")
        flags += ["-fsanitize=address", "-fno-omit-frame-pointer", "-g", "-shared-libsan"]
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(HERE, "hipemu.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    srcs.append(os.path.join(HERE, "hipemu_impl.cpp"))
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.splitext(os.path.basename(s))[0] + ".o")
        objs.append(o)
        newest = max(os.path.getmtime(p) for p in [s] + deps)
        if force or not os.path.exists(o) or os.path.getmtime(o) < newest:
            jobs.append([CXX, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[emu build]", cmd[-3], flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu compile failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([CXX, "-shared", "-fPIC", "-pthread", *(["-fsanitize=address", "-shared-libsan"] if asan else []), *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build_emu(verbose=True))
