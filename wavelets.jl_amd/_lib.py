"""ctypes binding of libwavelets_mi355x.so (include/wavelets_mi355x.h).

There is deliberately NO fallback: if the HIP library is missing or no gfx950 device is
present, importing the symbols works (so CPU-only tests can check the ABI) but creating a
context raises -- the product path never runs on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwavelets_mi355x.so")
# The opt-in fused build (same sources, `make -C wavelets.jl_amd/csrc FMA=1`: a*b+c may contract to one rounding).  "exact" is the
# default and the only mode whose results are bit-identical to the reference; "fused" agrees with it to the tolerances SURVEY.md
# 8(c) states (tests/test_gpu_fused.py).  Selected explicitly with transforms.set_arithmetic(); never a fallback for one another.
LIB_PATHS = {"exact": LIB_PATH, "fused": os.path.join(_HERE, "libwavelets_mi355x_fma.so")}

WL_F32, WL_F64 = 0, 1

STATUS = {
    0: "WL_OK", -1: "WL_EINVAL_SIZE", -2: "WL_EINVAL_L", -3: "WL_EALIAS", -4: "WL_EDIMS",
    -5: "WL_EINVAL_CUBE", -6: "WL_EINVAL_TREE", -7: "WL_EINVAL_SCHEME", -8: "WL_EINVAL_DTYPE",
    -9: "WL_EINVAL_FILTER", -10: "WL_EINVAL_ARG", -11: "WL_ENOMEM", -12: "WL_EHIP", -13: "WL_ENODEVICE",
}


class WaveletsLibraryError(RuntimeError):
    pass


_libs = {}
_mode = "exact"

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/wavelets_mi355x.h declares
SIGNATURES = {
    "wl_version": (C.c_int, []),
    "wl_strerror": (C.c_char_p, [C.c_int]),
    "wl_maxtransformlevels": (C.c_int, [C.c_int64]),
    "wl_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "wl_ctx_destroy": (C.c_int, [_vp]),
    "wl_shard_range": (C.c_int, [C.c_int64, C.c_int, C.c_int, _i64p, _i64p]),
    "wl_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, _i64p, C.c_int]),
    "wl_workspace_bytes_full": (C.c_size_t, [C.c_int, C.c_int, _i64p, C.c_int]),
    "wl_ctx_reserve": (C.c_int, [_vp, C.c_size_t]),
    "wl_ctx_workspace_held": (C.c_size_t, [_vp]),
    "wl_stream_sync": (C.c_int, [_vp, _vp]),
    "wl_last_hip_error": (C.c_int, [_vp]),
    "wl_ctx_set_path": (C.c_int, [_vp, C.c_int]),
    "wl_last_kernel": (C.c_char_p, [_vp]),
    "wl_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "wl_ctx_clear_options": (C.c_int, [_vp]),
    "wl_dwt_filter": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i64p, _f64p, C.c_int, C.c_int, C.c_int, _vp]),
    "wl_dwt_lifting": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _i64p, C.c_int, _i32p, _i32p, _i32p, _f64p,
                                 C.c_double, C.c_double, C.c_int, C.c_int, _vp]),
    "wl_dwt_lifting_oop": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i64p, C.c_int, _i32p, _i32p, _i32p, _f64p,
                                     C.c_double, C.c_double, C.c_int, C.c_int, _vp]),
    "wl_dwtc_filter": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, _f64p, C.c_int,
                                 C.c_int, C.c_int, _vp]),
    "wl_dwtc_lifting": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, _i32p, _i32p, _i32p,
                                  _f64p, C.c_double, C.c_double, C.c_int, C.c_int, _vp]),
    "wl_dwtc_lifting_oop": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, _i32p, _i32p, _i32p,
                                  _f64p, C.c_double, C.c_double, C.c_int, C.c_int, _vp]),
    "wl_dwt_filter_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _i64p, C.c_int64, C.c_int64, _f64p, C.c_int, C.c_int, C.c_int, _vp]),
    "wl_wpt_filter": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int64, _f64p, C.c_int, _u8p, C.c_int64, C.c_int, _vp]),
    "wl_wpt_lifting": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, C.c_int, _i32p, _i32p, _i32p, _f64p,
                                 C.c_double, C.c_double, _u8p, C.c_int64, C.c_int, _vp]),
    "wl_wpt_filter_full": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int64, _f64p, C.c_int, C.c_int, C.c_int, _vp]),
    "wl_wpt_lifting_full": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, C.c_int, _i32p, _i32p, _i32p, _f64p,
                                      C.c_double, C.c_double, C.c_int, C.c_int, _vp]),
    "wl_maxmodwttransformlevels": (C.c_int, [C.c_int64]),
    "wl_modwt": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, _vp, C.c_int64, _f64p, C.c_int, C.c_int, _vp]),
    "wl_imodwt": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int64, C.c_int64, C.c_int, _f64p, C.c_int, _vp]),
    "wl_threshold": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, C.c_int, C.c_double, C.c_int, _vp]),
    "wl_threshold_biggest": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, C.c_int64, _vp]),
    "wl_median": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, _f64p, _vp]),
    "wl_mad": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, _f64p, _vp]),
    "wl_denoise_ti_filter": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i64p, _f64p, C.c_int, C.c_int, C.c_int, C.c_double, _i64p,
                                       C.c_double, _vp]),
    "wl_denoise_ti_lifting": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i64p, C.c_int, _i32p, _i32p, _i32p, _f64p, C.c_double, C.c_double,
                                        C.c_int, C.c_int, C.c_double, _i64p, C.c_double, _vp]),
    "wl_circshift": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _i64p, _i64p, _vp]),
    "wl_arrayadd": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int64, _vp]),
    "wl_rmul": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, C.c_double, _vp]),
}


def load(mode=None):
    """Load the shared library of `mode` (default: the selected one; no GPU needed for this step) and bind every ABI symbol."""
    mode = _mode if mode is None else mode
    lib = _libs.get(mode)
    if lib is not None:
        return lib
    path = LIB_PATHS[mode]
    if not os.path.exists(path):
        raise WaveletsLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C wavelets.jl_amd/csrc{' FMA=1' if mode == 'fused' else ''}`).  There is no CPU fallback.")
    lib = C.CDLL(path)
    try:
        for nm, (res, args) in SIGNATURES.items():
            fn = getattr(lib, nm)          # AttributeError if the symbol is missing -> loud
            fn.restype = res
            fn.argtypes = args
    except AttributeError as e:
        raise WaveletsLibraryError(f"{path} is out of date (no symbol {e}): rebuild it") from e
    _libs[mode] = lib
    return lib


def arithmetic() -> str:
    return _mode


def _select(mode: str):
    """transforms.set_arithmetic() is the public switch (it also retires the contexts of the library being left)."""
    global _mode
    if mode not in LIB_PATHS:
        raise ValueError(f"arithmetic mode must be one of {sorted(LIB_PATHS)}, got {mode!r}")
    _mode = mode


def strerror(rc: int) -> str:
    return load().wl_strerror(rc).decode()
