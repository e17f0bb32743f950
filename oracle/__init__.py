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
       -9: "bad filter", -10: "bad threshold"}


class OracleError(ValueError):
    def __init__(self, rc):
        super().__init__(f"oracle status {rc}: {ERR.get(rc, '?')}")
        self.rc = rc


def build(force: bool = False):
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(LIB_PATH)
            for f in ("wl_oracle.c", "wl_oracle_impl.h", "wl_oracle_ext.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def build_native(out_dir: str | None = None) -> C.CDLL:
    """CPU-BASELINE build (bench.py's cpu_baseline leg only): the same sources compiled `-O3 -march=native
    -ffp-contract=off` ON THE HOST THAT RUNS THE BENCH, into a scratch directory -- never the
    library the parity tests use (that one stays -O2 without -march so that it can travel between machines)."""
    import tempfile
    out_dir = out_dir or os.path.join(tempfile.gettempdir(), "wl_oracle_native_%d" % os.getuid())
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libwl_oracle_native.so")
    srcs = [os.path.join(_HERE, f) for f in ("wl_oracle.c", "wl_oracle_ext.c")]
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("wl_oracle.c", "wl_oracle_impl.h", "wl_oracle_ext.c"))
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                               "-fvisibility=hidden", "-std=c11", "-fopenmp", "-shared", "-o", so] + srcs + ["-lm"])
    return C.CDLL(so)


def time_dwt_filter(handle: C.CDLL, xf: np.ndarray, qmf, L: int, reps: int = 3, warmup: int = 1, threads: int = 1):
    """Wall-clock seconds of each of `reps` calls of the C entry point itself (no Python-side copies inside the timed region):
    `xf` must be Fortran-ordered (Julia layout).  threads > 1 uses the OpenMP-over-lines variant (2-D only).  Returns the
    list of per-call seconds after `warmup` untimed calls."""
    import time
    assert xf.flags.f_contiguous
    yf = np.empty_like(xf, order="F")
    q = np.ascontiguousarray(qmf, dtype=np.float64)
    qp = q.ctypes.data_as(C.POINTER(C.c_double))
    if threads > 1:
        assert xf.ndim == 2
        call = lambda: handle.wlo_dwt2d_filter_mt(_dt(xf), _p(yf), _p(xf), C.c_int64(xf.shape[0]), C.c_int64(xf.shape[1]), qp, len(q), int(L), 1)
    else:
        call = lambda: handle.wlo_dwt_filter(_dt(xf), _p(yf), _p(xf), xf.ndim, _dims(xf), qp, len(q), int(L), 1)
    out = []
    for i in range(warmup + reps):
        t0 = time.perf_counter()
        rc = call()
        dt = time.perf_counter() - t0
        if rc:
            raise OracleError(rc)
        if i >= warmup:
            out.append(dt)
    return out


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


def dwt2d_filter_mt(x: np.ndarray, qmf, L=None, fw=True) -> np.ndarray:
    """2-D filter dwt / idwt, the per-level line loops on all OpenMP threads (bit-identical to dwt_filter)"""
    xf = _col(x)
    yf = np.empty_like(xf, order="F")
    q = np.ascontiguousarray(qmf, dtype=np.float64)
    L = maxtransformlevels(x) if L is None else L
    rc = lib().wlo_dwt2d_filter_mt(_dt(xf), _p(yf), _p(xf), C.c_int64(x.shape[0]), C.c_int64(x.shape[1]),
                                   q.ctypes.data_as(C.POINTER(C.c_double)), len(q), int(L), 1 if fw else 0)
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


# ---- section 8(f) rows 3-4: modwt, threshold, noise estimate (wl_oracle_ext.c; parity unpinned) ----
TH_KINDS = {"hard": 0, "soft": 1, "semisoft": 2, "stein": 3, "pos": 4, "neg": 5}


def maxmodwttransformlevels(n) -> int:
    n = int(n.shape[0]) if hasattr(n, "shape") else int(n)
    return int(lib().wlo_maxmodwttransformlevels(C.c_int64(n)))


def modwt(x: np.ndarray, qmf, L=None) -> np.ndarray:
    """modwt(x, wt, L) -> logical (N, L+1) array (transforms_maximal_overlap.jl:47-63)"""
    xf = np.ascontiguousarray(x)
    N = len(xf)
    L = maxmodwttransformlevels(N) if L is None else int(L)
    q = np.ascontiguousarray(qmf, dtype=np.float64)
    out = np.empty((max(L, 0) + 1, N), dtype=xf.dtype)          # column-major N x (L+1)
    rc = lib().wlo_modwt(_dt(xf), _p(out), _p(xf), C.c_int64(N), q.ctypes.data_as(C.POINTER(C.c_double)), len(q), L)
    if rc:
        raise OracleError(rc)
    return np.ascontiguousarray(out.T)


def imodwt(xw: np.ndarray, qmf) -> np.ndarray:
    N, ncols = xw.shape
    xf = np.ascontiguousarray(xw.T)                              # column-major N x ncols
    q = np.ascontiguousarray(qmf, dtype=np.float64)
    out = np.empty(N, dtype=xw.dtype)
    rc = lib().wlo_imodwt(_dt(xf), _p(out), _p(xf), C.c_int64(N), int(ncols), q.ctypes.data_as(C.POINTER(C.c_double)), len(q))
    if rc:
        raise OracleError(rc)
    return out


def threshold(x: np.ndarray, kind: str, t=None, m=None) -> np.ndarray:
    """threshold(x, TH, t): t a Python int -> arithmetic in the element type, a float -> Julia Float64 promotion."""
    y = _col(x) if x.ndim > 1 else np.ascontiguousarray(x).copy()
    if kind == "biggest":
        rc = lib().wlo_threshold_biggest(_dt(y), _p(y), C.c_int64(y.size), C.c_int64(int(m)))
    else:
        t_is_f64 = 0 if (t is None or isinstance(t, (int, np.integer)) or isinstance(t, np.floating) and t.dtype == y.dtype) else 1
        rc = lib().wlo_threshold(_dt(y), _p(y), C.c_int64(y.size), TH_KINDS[kind], C.c_double(0.0 if t is None else float(t)), t_is_f64)
    if rc:
        raise OracleError(rc)
    return np.ascontiguousarray(y)


def median(v: np.ndarray) -> float:
    w = np.ascontiguousarray(v).copy()
    out = C.c_double()
    rc = lib().wlo_median(_dt(w), _p(w), C.c_int64(w.size), C.byref(out))
    if rc:
        raise OracleError(rc)
    return out.value


def mad(v: np.ndarray) -> float:
    w = np.ascontiguousarray(v).copy()
    out = C.c_double()
    rc = lib().wlo_mad(_dt(w), _p(w), C.c_int64(w.size), C.byref(out))
    if rc:
        raise OracleError(rc)
    return out.value


def circshift(a: np.ndarray, shift) -> np.ndarray:
    af = _col(a)
    bf = np.empty_like(af, order="F")
    sh = [int(s) for s in (shift if hasattr(shift, "__len__") else [shift])]
    rc = lib().wlo_circshift(_dt(af), _p(bf), _p(af), a.ndim, _dims(a), (C.c_int64 * 3)(*(sh + [0] * (3 - len(sh)))))
    if rc:
        raise OracleError(rc)
    return np.ascontiguousarray(bf)


def arrayadd(y: np.ndarray, z: np.ndarray) -> np.ndarray:
    yy = np.ascontiguousarray(y).copy()
    zz = np.ascontiguousarray(z)
    rc = lib().wlo_arrayadd(_dt(yy), _p(yy), _p(zz), C.c_int64(yy.size))
    if rc:
        raise OracleError(rc)
    return yy


def rmul(y: np.ndarray, s: float) -> np.ndarray:
    yy = np.ascontiguousarray(y).copy()
    rc = lib().wlo_rmul(_dt(yy), _p(yy), C.c_int64(yy.size), C.c_double(float(s)))
    if rc:
        raise OracleError(rc)
    return yy


def noisest(x: np.ndarray, transform, L=1) -> float:
    """noisest (denoising.jl:92-101): `transform(x, L)` is the oracle dwt for the wavelet (None: identity);
    y[detailrange(y, L)] is LINEAR indexing with size(y, 1): for matrices it takes the lower half of the first column."""
    y = x if transform is None else transform(x, L)
    n1 = y.shape[0]
    # detailrange(n, L) = round(Int, n/2^L + 1) : round(Int, n/2^(L-1)), ties to even (non_dyadic.jl:7); Python's round()
    # is the same banker's rounding.  0-based half-open: [lo, hi)
    lo, hi = int(round(n1 / 2 ** L + 1)) - 1, int(round(n1 / 2 ** (L - 1)))
    dr = np.asfortranarray(y).reshape(-1, order="F")[lo:hi].copy()
    return mad(dr) / 0.6745


def denoise(x: np.ndarray, fwd, inv, L, th_kind, t_unit, TI=False, nspin=None, lifting=False, noise_transform="same", sigma=None):
    """denoise (denoising.jl:21-81) composed from the oracle pieces in the reference's order.
    fwd(a, L) / inv(a, L): oracle dwt / idwt for the wavelet (None: wt === nothing); t_unit = dnt.t; sigma: the value a
    custom `estnoise` returned (default: noisest, denoising.jl:33)"""
    if sigma is None:
        sigma = noisest(x, (lambda a, l: fwd(a, l)) if fwd is not None else None)
    t = sigma * t_unit
    if TI:
        nsp = tuple(nspin) if hasattr(nspin, "__len__") else (int(nspin),)
        pns = int(np.prod(nsp))
        y = np.zeros_like(x)
        for i in range(pns):
            if x.ndim == 1:
                shift = [i]                              # denoising.jl:40-42: vectors are shifted by i - 1 whatever nspin's length
            else:
                rem, shift = i, []
                for d in nsp:                            # CartesianIndices: first dimension fastest (nspin2circ, :113-121)
                    shift.append(rem % d)
                    rem //= d
            shift = shift + [0] * (x.ndim - len(shift))
            z = circshift(x, shift)
            xt = threshold(fwd(z, L), th_kind, t)
            z = inv(xt, L)
            z = circshift(z, [-s for s in shift])
            y = arrayadd(y, z).reshape(x.shape)
        return rmul(y, 1 / pns).reshape(x.shape)
    if fwd is None:
        return threshold(x, th_kind, t)
    return inv(threshold(fwd(x, L), th_kind, t), L)
