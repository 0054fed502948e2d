"""The gfx950 library must load (no GPU needed) and export every symbol include/stx_hip.h declares,
and the Python binding table must cover exactly that set."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "stx_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(stx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_declared_symbols():
    from stereo_toolbox_amd.build import build_hip
    lib = ctypes.CDLL(build_hip(verbose=False))
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libstx_hip.so does not export {s}"
    assert b"gfx950" in ctypes.cast(lib.stx_build_info, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_binding_table_matches_header():
    from stereo_toolbox_amd._capi import SIGNATURES
    syms = set(declared_symbols()) - {"stx_last_error", "stx_build_info"}
    assert syms == set(SIGNATURES), (syms ^ set(SIGNATURES))


def test_emulator_build_exports_same_abi(emu):
    for s in declared_symbols():
        assert hasattr(emu._dll, s), s
    assert "hipemu" in emu.build_info()
