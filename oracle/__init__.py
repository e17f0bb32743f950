"""oracle -- ctypes wrapper of oracle/libwl_oracle.so.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The library is a literal C restatement of the reference's CPU loops (see wl_oracle.c); numpy
arrays passed here are interpreted in Julia layout by *logical index*: a numpy array `a` of
shape (m, n) stands for the Julia matrix A with A[i+1, j+1] == a[i, j] (the wrapper copies to
and from column-major storage).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwl_oracle.so")
_lib = None

ERR = {-1: "size must have a sufficient power of 2 factor", -2: "L must be positive", -3: "in array is out array",
       -4: "bad dims", -5: "array must be square/cube", -6: "invalid tree", -7: "bad scheme", -8: "bad dtype",
       -9: "bad filter"}


class OracleError(ValueError):
    def __init__(self, rc):
        super().__init__(f"oracle status {rc}: {ERR.get(rc, '?')}")
        self.rc = rc


def build(force: bool = False):
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(LIB_PATH)
            for f in ("wl_oracle.c", "wl_oracle_impl.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _dt(a):
    if a.dtype == np.float32:
        return 0
    if a.dtype == np.float64:
        return 1
    raise TypeError(a.dtype)


def _col(a):
    """logical array -> Fortran-ordered contiguous copy"""
    return np.asfortranarray(a).copy(order="F")


def _dims(a):
    return (C.c_int64 * 3)(*([int(s) for s in a.shape] + [1] * (3 - a.ndim)))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _scheme_args(scheme):
    iu, nc, sh, cf = scheme.flatten()
    return (len(iu), iu.ctypes.data_as(C.POINTER(C.c_int32)), nc.ctypes.data_as(C.POINTER(C.c_int32)),
            sh.ctypes.data_as(C.POINTER(C.c_int32)), cf.ctypes.data_as(C.POINTER(C.c_double)),
            C.c_double(scheme.norm1), C.c_double(scheme.norm2)), (iu, nc, sh, cf)


def maxtransformlevels(x):
    if hasattr(x, "shape"):
        return min(lib().wlo_maxtransformlevels_n(C.c_int64(int(s))) for s in x.shape)
    return lib().wlo_maxtransformlevels_n(C.c_int64(int(x)))


def dwt_filter(x: np.ndarray, qmf, L=None, fw=True) -> np.ndarray:
    """dwt/idwt with OrthoFilter taps `qmf` (Float64), reference _dwt! (transforms_filter.jl:13-294)."""
    xf = _col(x)
    yf = np.empty_like(xf, order="F")
    q = np.ascontiguousarray(qmf, dtype=np.float64)
    L = maxtransformlevels(x) if L is None else L
    rc = lib().wlo_dwt_filter(_dt(xf), _p(yf), _p(xf), x.ndim, _dims(x), q.ctypes.data_as(C.POINTER(C.c_double)),
                              len(q), int(L), 1 if fw else 0)
    if rc:
        raise OracleError(rc)
    return np.ascontiguousarray(yf)


def dwt2d_filter_mt(x: np.ndarray, qmf, L=None) -> np.ndarray:
    """forward 2-D filter dwt, the per-level line loops on all OpenMP threads (same results as dwt_filter)"""
    xf = _col(x)
    yf = np.empty_like(xf, order="F")
    q = np.ascontiguousarray(qmf, dtype=np.float64)
    L = maxtransformlevels(x) if L is None else L
    rc = lib().wlo_dwt2d_filter_mt(_dt(xf), _p(yf), _p(xf), C.c_int64(x.shape[0]), C.c_int64(x.shape[1]),
                                   q.ctypes.data_as(C.POINTER(C.c_double)), len(q), int(L))
    if rc:
        raise OracleError(rc)
    return np.ascontiguousarray(yf)


def max_threads() -> int:
    return int(lib().wlo_max_threads())


def dwt_lifting(x: np.ndarray, scheme, L=None, fw=True) -> np.ndarray:
    """dwt/idwt with a GLS scheme (copy + in place), reference _dwt! (transforms_lifting.jl:30-278)."""
    yf = _col(x)
    L = maxtransformlevels(x) if L is None else L
    args, keep = _scheme_args(scheme)
    rc = lib().wlo_dwt_lifting(_dt(yf), _p(yf), x.ndim, _dims(x), *args, int(L), 1 if fw else 0)
    if rc:
        raise OracleError(rc)
    return np.ascontiguousarray(yf)


def dwtc_filter(x: np.ndarray, qmf, L=None, fw=True) -> np.ndarray:
    xf = _col(x)
    yf = np.empty_like(xf, order="F")
    q = np.ascontiguousarray(qmf, dtype=np.float64)
    n, ns = x.shape
    L = maxtransformlevels(n) if L is None else L
    rc = lib().wlo_dwtc_filter(_dt(xf), _p(yf), _p(xf), C.c_int64(n), C.c_int64(ns), C.c_int64(n),
                               q.ctypes.data_as(C.POINTER(C.c_double)), len(q), int(L), 1 if fw else 0)
    if rc:
        raise OracleError(rc)
    return np.ascontiguousarray(yf)


def dwtc_lifting(x: np.ndarray, scheme, L=None, fw=True) -> np.ndarray:
    yf = _col(x)
    n, ns = x.shape
    L = maxtransformlevels(n) if L is None else L
    args, keep = _scheme_args(scheme)
    rc = lib().wlo_dwtc_lifting(_dt(yf), _p(yf), C.c_int64(n), C.c_int64(ns), C.c_int64(n), *args, int(L),
                                1 if fw else 0)
    if rc:
        raise OracleError(rc)
    return np.ascontiguousarray(yf)


def wpt_filter(x: np.ndarray, qmf, tree, fw=True) -> np.ndarray:
    xf = np.ascontiguousarray(x)
    yf = np.empty_like(xf)
    q = np.ascontiguousarray(qmf, dtype=np.float64)
    t = np.ascontiguousarray(np.asarray(tree).astype(np.uint8))
    rc = lib().wlo_wpt_filter(_dt(xf), _p(yf), _p(xf), C.c_int64(len(xf)), q.ctypes.data_as(C.POINTER(C.c_double)),
                              len(q), t.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(t)), 1 if fw else 0)
    if rc:
        raise OracleError(rc)
    return yf


def wpt_lifting(x: np.ndarray, scheme, tree, fw=True) -> np.ndarray:
    yf = np.ascontiguousarray(x).copy()
    t = np.ascontiguousarray(np.asarray(tree).astype(np.uint8))
    args, keep = _scheme_args(scheme)
    rc = lib().wlo_wpt_lifting(_dt(yf), _p(yf), C.c_int64(len(yf)), *args,
                               t.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(t)), 1 if fw else 0)
    if rc:
        raise OracleError(rc)
    return yf


# ---- building blocks (1-based index arguments exactly as the reference passes them) ----------
def filtdown(f, out, iout, nout, x, ix, shift=0, ss=False):
    f = np.ascontiguousarray(f)
    rc = lib().wlo_filtdown(_dt(f), _p(f), len(f), _p(out), C.c_int64(iout), C.c_int64(nout), _p(x),
                            C.c_int64(ix), C.c_int64(shift), 1 if ss else 0)
    if rc:
        raise OracleError(rc)


def filtup(add2out, f, out, iout, nout, x, ix, shift=0, ss=False):
    f = np.ascontiguousarray(f)
    rc = lib().wlo_filtup(_dt(f), 1 if add2out else 0, _p(f), len(f), _p(out), C.c_int64(iout), C.c_int64(nout),
                          _p(x), C.c_int64(ix), C.c_int64(shift), 1 if ss else 0)
    if rc:
        raise OracleError(rc)


def makereverseqmfpair(qmf, fw, dtype):
    q = np.ascontiguousarray(qmf, dtype=np.float64)
    s = np.empty(len(q), dtype=dtype)
    d = np.empty(len(q), dtype=dtype)
    rc = lib().wlo_makereverseqmfpair(_dt(s), q.ctypes.data_as(C.POINTER(C.c_double)), len(q), 1 if fw else 0,
                                      _p(s), _p(d))
    if rc:
        raise OracleError(rc)
    return s, d


def lift(x, half, is_update, shift, coef):
    coef = np.ascontiguousarray(coef, dtype=x.dtype)
    rc = lib().wlo_lift(_dt(x), _p(x), C.c_int64(half), 1 if is_update else 0, len(coef), int(shift), _p(coef))
    if rc:
        raise OracleError(rc)


def split(a):
    lib().wlo_split(_dt(a), _p(a), C.c_int64(len(a)))
    return a


def merge(a):
    lib().wlo_merge(_dt(a), _p(a), C.c_int64(len(a)))
    return a
