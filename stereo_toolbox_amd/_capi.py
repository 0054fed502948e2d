"""ctypes binding of the C-ABI declared in include/stx_hip.h.

The product path loads exactly one library: the in-tree gfx950 build
`stereo_toolbox_amd/lib/libstx_hip.so`.  There is no CPU fallback: if the library is missing or a
tensor is not on a ROCm device the call raises.  (tests/ may instantiate `StxLib` on the host
emulator build of the same sources to check index math without a GPU -- see tests/hipemu.)
"""
import ctypes
import os
import threading

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be the one already
# mapped when libstx_hip.so is dlopen'ed, otherwise the loader pulls a second runtime from /opt/rocm
# and launches fail with "no ROCm-capable device is detected".
import torch  # noqa: F401  (side effect: loads torch's HIP runtime first)

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_longlong

# name -> argtypes (every entry point returns int: 0 ok, else see stx_last_error()).
SIGNATURES = {
    "stx_get_tuning": [ctypes.c_char_p],
    "stx_set_tuning": [ctypes.c_char_p, _I],
    # cost_volume.hip
    "stx_cost_volume_fwd": [_P, _P, _I, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "stx_cost_volume_bwd": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    # group_normalize.hip
    "stx_group_normalize_fwd": [_P, _P, _I, _I, _I, _I, _F, _P],
    "stx_group_normalize_bwd": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    # head.hip
    "stx_head_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "stx_head_bwd_workspace_floats": [_I, _I, _I, _I],
    "stx_head_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "stx_head_fwd2": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "stx_head_bwd2": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "stx_softargmax_fwd": [_P, _P, _I, _I, _I, _P],
    "stx_argmax_fwd": [_P, _P, _I, _I, _I, _P],
    "stx_softmax_d_fwd": [_P, _P, _I, _I, _I, _P],
    # estimators.hip
    "stx_unimodal_fwd": [_P, _P, _I, _I, _I, _P],
    "stx_dominant_modal_fwd": [_P, _P, _I, _I, _I, _P],
    "stx_modal_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "stx_modal_bwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "stx_split_mode": [_P, _P, _P, _I, _I, _I, _P],
    # conv3d.hip
    "stx_conv3d_packed_floats": [_I, _I, _I],
    "stx_conv3d_pack_weight": [_P, _P, _I, _I, _I, _I, _P],
    "stx_conv3d_fwd_blocks": [_I, _I, _I],
    "stx_conv3d_fwd_stat_rows": [_I, _I, _I, _I, _I, _I, _I, _I],
    "stx_deconv3d_fwd_blocks": [_I, _I, _I],
    "stx_conv3d_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "stx_deconv3d_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "stx_conv3d_wgrad_workspace_floats": [_I, _I, _I, _I, _I, _I, _I, _I],
    "stx_conv3d_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "stx_conv3d_wgrad_bn_supported": [_I, _I, _I, _I, _I, _I],
    "stx_conv3d_wgrad_bn": [_P] * 9 + [_F, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    # conv2d.hip
    "stx_conv2d_supported": [_I, _I],
    "stx_conv2d_stat_rows": [_I],
    "stx_conv2d_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    # conv_c1.hip
    "stx_conv3d_c1_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "stx_conv3d_c1_wgrad_workspace_floats": [_I],
    "stx_conv3d_c1_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "stx_conv3d_c1_dgrad": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    # act.hip
    "stx_mish_fwd": [_P, _P, _L, _P],
    "stx_mish_bwd": [_P, _P, _P, _L, _P],
    "stx_depth_to_space": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "stx_gate_fwd": [_P, _P, _P, _I, _I, _L, _I, _P],
    "stx_gate_bwd": [_P, _P, _P, _P, _P, _I, _I, _L, _I, _P],
    "stx_concat_channels": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _L, _P],
    "stx_split_channels": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _P],
    "stx_transpose": [_P, _P, _I, _I, _I, _P],
    # acv.hip
    "stx_dwconv_hw_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "stx_dwconv_hw_wgrad_workspace_floats": [_I],
    "stx_dwconv_hw_wgrad": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "stx_cost_volume_scale_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "stx_scale_channels": [_P, _P, _P, _L, _I, _P],
    "stx_ac_volume_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    # preprocess.hip
    "stx_pad_normalize_u8": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    # refine2d.hip
    "stx_warp_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "stx_warp_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "stx_corr_volume_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "stx_corr_volume_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "stx_disparity_variance_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "stx_disparity_variance_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "stx_sampled_volume_fwd": [_P, _P, _I, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "stx_sampled_volume_bwd": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    # bn.hip
    "stx_bn_reduce_blocks": [],
    "stx_bn_stats_rows": [_L, _I],
    "stx_bn_stats": [_P, _P, _L, _I, _I, _P],
    "stx_bn_finalize": [_P, _I, _I, ctypes.c_double, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P],
    "stx_bn_finalize_groups": [_P, _I, _I, ctypes.c_double, _P, _P, _P, _P, _F, _F, _P, _I, _P],
    "stx_bn_apply": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "stx_bn_bwd_reduce": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "stx_bn_bwd_apply": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "stx_bn_bwd_reduce2": [_P] * 14 + [_L, _I, _I, _I, _P],
    "stx_bn_bwd_apply2": [_P] * 18 + [_L, _I, _I, _I, _P],
}
_RET_CHARP = ("stx_last_error", "stx_build_info")


class StxError(RuntimeError):
    pass


class StxLib:
    """Thin typed handle over one shared object exporting the stx_* C-ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise StxError(
                f"HIP library not found: {path}. Build it with `python -m stereo_toolbox_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback for the cost-volume hot path.")
        self.path = path
        self._dll = ctypes.CDLL(path)
        for name in _RET_CHARP:
            fn = getattr(self._dll, name)
            fn.restype = ctypes.c_char_p
            fn.argtypes = []
        self._fns = {}
        for name, argtypes in SIGNATURES.items():
            fn = getattr(self._dll, name, None)
            if fn is None:
                continue
            fn.restype = ctypes.c_longlong if name.endswith(("_floats", "_stat_rows")) else _I
            fn.argtypes = argtypes
            self._fns[name] = fn

    def has(self, name):
        return name in self._fns

    def build_info(self):
        return self._dll.stx_build_info().decode()

    def get_tuning(self, name):
        return self._fns["stx_get_tuning"](name.encode())

    def set_tuning(self, name, value):
        """A/B switch of the library (see StxTune in csrc/stx_common.h); returns the previous value."""
        old = self.get_tuning(name)
        if old < 0 or self._fns["stx_set_tuning"](name.encode(), int(value)) != 0:
            raise StxError(f"unknown tuning switch {name}")
        return old

    def raw(self, name):
        return self._fns[name]

    def call(self, name, *args):
        fn = self._fns.get(name)
        if fn is None:
            raise StxError(f"{self.path} does not export {name}")
        rc = fn(*args)
        if rc != 0:
            raise StxError(f"{name} failed ({rc}): {self._dll.stx_last_error().decode()}")


_LIB = None
_LOCK = threading.Lock()
# STX_HIP_LIB: another build of the SAME sources (e.g. tools/build_variant.sh -DSTX_PRECISE_MATH) for A/B measurements; it
# must exist -- a missing file raises, nothing falls back.
LIB_PATH = os.environ.get("STX_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libstx_hip.so")


def get_lib():
    """The gfx950 library (loaded once per process). Raises StxError if it has not been built."""
    global _LIB
    if _LIB is None:
        with _LOCK:
            if _LIB is None:
                _LIB = StxLib(LIB_PATH)
    return _LIB
