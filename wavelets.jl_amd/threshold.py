"""Threshold -- host-side mirror of src/Threshold/threshold_main.jl (threshold!/threshold, the THType
singletons) and src/Threshold/denoising.jl (VisuShrink, denoise, noisest, mad!), the main in-package
caller of the transform path.  Julia's `f!` is spelled `f_`.

Everything runs on the device through libwavelets_mi355x.so (wl_threshold, wl_threshold_biggest,
wl_mad, wl_circshift, wl_arrayadd, wl_rmul + the dwt/idwt entry points); the only host scalars are
the ones the reference also has on the host (the noise estimate sigma and the threshold t).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from . import util as Util
from . import wt as WT
from .transforms import (ArgumentError, DimensionMismatch, HIPError, _check, _context, _dims, _dtype_code, _prep_in, _i32p, _f64p,
                         dwt, dwt_oop_, idwt, idwt_, idwt_oop_, is_julia_layout, julia_layout, similar)
from .wt import GLS, OrthoFilter, wavelet


# ---- threshold types (threshold_main.jl:8-17) --------------------------------------------------
class THType:
    code = None

    def __repr__(self):
        return type(self).__name__ + "()"


class HardTH(THType):
    code = 0


class SoftTH(THType):
    code = 1


class SemiSoftTH(THType):
    code = 2


class SteinTH(THType):
    code = 3


class PosTH(THType):
    code = 4


class NegTH(THType):
    code = 5


class BiggestTH(THType):
    code = -1


DEFAULT_TH = HardTH()


def _dev_array(x) -> torch.Tensor:
    if not isinstance(x, torch.Tensor) or x.device.type != "cuda":
        raise HIPError("expected a torch tensor resident on an MI355X device (use to_device(array)); there is no CPU path")
    _dtype_code(x)
    if not (x.is_contiguous() or is_julia_layout(x)):
        raise ArgumentError("array must be dense (Julia layout or C-contiguous)")
    return x


def _t_is_f64(x: torch.Tensor, t) -> int:
    """the type Julia's promotion computes `x[i] op t` in: Float64 for a Python float / np.float64 threshold,
    the element type for integers and for thresholds that already have the element type"""
    if isinstance(t, (bool, int, np.integer)):
        return 0
    if isinstance(t, np.floating):
        return 0 if (t.dtype == np.float32 and x.dtype == torch.float32) else 1
    return 1


def threshold_(x, TH: THType, t=None):
    """threshold!(x, TH, t) / threshold!(x, BiggestTH(), m) / threshold!(x, PosTH()|NegTH()) -- in place"""
    x = _dev_array(x)
    lib = _lib.load()
    h, st = _context(x.device)
    n = int(x.numel())
    if isinstance(TH, BiggestTH):
        if t is None or int(t) != t:
            raise TypeError("threshold!(x, BiggestTH(), m::Int)")
        if int(t) < 0:
            raise AssertionError("m >= 0")
        _check(lib.wl_threshold_biggest(h, _dtype_code(x), C.c_void_p(x.data_ptr()), n, int(t), st), h)
        return x
    if isinstance(TH, (PosTH, NegTH)):
        if t is not None:
            raise TypeError("threshold!(x, PosTH()/NegTH()) takes no threshold (MethodError in the reference)")
        _check(lib.wl_threshold(h, _dtype_code(x), C.c_void_p(x.data_ptr()), n, TH.code, 0.0, 0, st), h)
        return x
    if not isinstance(TH, THType) or TH.code is None:
        raise TypeError("unknown threshold type")
    if t is None:
        raise TypeError("threshold!(x, TH, t): t is required (MethodError in the reference)")
    if not float(t) >= 0:
        raise AssertionError("t >= 0")
    _check(lib.wl_threshold(h, _dtype_code(x), C.c_void_p(x.data_ptr()), n, TH.code, float(t), _t_is_f64(x, t), st), h)
    return x


def threshold(x, TH: THType, t=None):
    """threshold(x, TH[, t]): the non-in-place form (threshold_main.jl:120-127)"""
    x = _dev_array(x)
    y = similar(x)
    y.copy_(x)
    return threshold_(y, TH, t)


# ---- denoising (denoising.jl) -------------------------------------------------------------------
class DNFT:
    pass


class VisuShrink(DNFT):
    """VisuShrink(th, t) / VisuShrink(n): t = sqrt(2*log(n)) for noise level sigma = 1"""

    def __init__(self, *args):
        if len(args) == 1:
            self.th, self.t = DEFAULT_TH, math.sqrt(2 * math.log(int(args[0])))
        elif len(args) == 2:
            self.th, self.t = args[0], float(args[1])
        else:
            raise TypeError("VisuShrink(th, t) or VisuShrink(n)")


DEFAULT_WAVELET = wavelet(WT.sym5, WT.Filter)
_DEFAULT = object()


def mad_(y) -> float:
    """mad!(y): median absolute deviation; y is overwritten by abs.(y .- median(y)) (denoising.jl:103-110)"""
    y = _dev_array(y)
    h, st = _context(y.device)
    out = C.c_double()
    _check(_lib.load().wl_mad(h, _dtype_code(y), C.c_void_p(y.data_ptr()), int(y.numel()), C.byref(out), st), h)
    return out.value


def median(v) -> float:
    v = _dev_array(v)
    h, st = _context(v.device)
    out = C.c_double()
    vv = v if v.is_contiguous() else julia_layout(v)
    _check(_lib.load().wl_median(h, _dtype_code(vv), C.c_void_p(vv.data_ptr()), int(vv.numel()), C.byref(out), st), h)
    return out.value


def noisest(x, wt=_DEFAULT, L: int = 1) -> float:
    """noisest(x, wt=DEFAULT_WAVELET, L=1) (denoising.jl:92-101): MAD of the level-L detail range / 0.6745.
    `y[detailrange(y, L)]` is linear indexing with size(y, 1), as in the reference."""
    wt = DEFAULT_WAVELET if wt is _DEFAULT else wt
    x = _prep_in(x)
    y = x if wt is None else dwt(x, wt, L)
    n1 = int(y.shape[0])
    r = Util.detailrange(n1, L)            # round(Int, n/2^L + 1) : round(Int, n/2^(L-1)), ties to even (non_dyadic.jl:7)
    flat = y.t().reshape(-1) if y.dim() == 2 else (y.permute(2, 1, 0).reshape(-1) if y.dim() == 3 else y)
    dr = flat[r.start - 1:r.stop - 1].clone()
    return mad_(dr) / 0.6745


def nspin2circ(nspin, i: int):
    """shift vector of spin number i (1-based), first dimension fastest (denoising.jl:112-121)"""
    nsp = (int(nspin),) if not hasattr(nspin, "__len__") else tuple(int(s) for s in nspin)
    rem, c = int(i) - 1, []
    for d in nsp:
        c.append(rem % d)
        rem //= d
    return c


def circshift(a, shift) -> torch.Tensor:
    """b[i] = a[i - shift] along every dimension (Util.circshift! / Base.circshift)"""
    a = _prep_in(a)
    sh = [int(shift)] if not hasattr(shift, "__len__") else [int(s) for s in shift]
    if len(sh) > a.dim():
        raise DimensionMismatch("more shifts than dimensions")
    sh = sh + [0] * (3 - len(sh))
    b = similar(a)
    h, st = _context(a.device)
    _check(_lib.load().wl_circshift(h, _dtype_code(a), C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), a.dim(),
                                    _dims(a), (C.c_int64 * 3)(*sh), st), h)
    return b


def _arrayadd_(y, z):
    if y.numel() != z.numel():
        raise DimensionMismatch("lengths must be equal")
    h, st = _context(y.device)
    _check(_lib.load().wl_arrayadd(h, _dtype_code(y), C.c_void_p(y.data_ptr()), C.c_void_p(z.data_ptr()), int(y.numel()), st), h)
    return y


def _rmul_(y, s: float):
    h, st = _context(y.device)
    _check(_lib.load().wl_rmul(h, _dtype_code(y), C.c_void_p(y.data_ptr()), int(y.numel()), float(s), st), h)
    return y


def denoise(x, wt=_DEFAULT, L: Optional[int] = None, dnt: Optional[DNFT] = None, estnoise=noisest, TI: bool = False,
            nspin: Union[int, Sequence[int], None] = None) -> torch.Tensor:
    """denoise(x, wt=DEFAULT_WAVELET; L=min(maxtransformlevels(x),6), dnt=VisuShrink(size(x,1)), estnoise=noisest,
    TI=false, nspin=8 per dimension) -- denoising.jl:21-81.  wt=None is the reference's `nothing`."""
    wt = DEFAULT_WAVELET if wt is _DEFAULT else wt
    x = _prep_in(x)
    L = min(Util.maxtransformlevels(x), 6) if L is None else int(L)
    dnt = VisuShrink(int(x.shape[0])) if dnt is None else dnt
    nspin = tuple(8 for _ in range(x.dim())) if nspin is None else nspin
    if not Util.iscube(x):
        raise ArgumentError("array must be square/cube")
    nsp = (int(nspin),) if not hasattr(nspin, "__len__") else tuple(int(s) for s in nspin)
    # fast path only where it is the reference's own branch: threshold!(xt, th, t) exists for Hard / Soft / Semisoft / Stein
    # (codes 0..3; Pos / Neg take no t and raise in the loop below exactly as the reference's MethodError does); matrices need
    # one nspin entry per dimension (anything else goes through nspin2circ / circshift in the loop below, as in the reference)
    # The plain (not translation-invariant) denoise of an orthogonal filter takes the same device-resident call with ONE spin
    # of shift zero: y = (0 + idwt(threshold!(dwt(x)))) * 1 -- the same bits (a -0.0 may come back as +0.0), and the noise
    # estimate never leaves the device (the reference's order needs sigma on the host between noisest and threshold!).
    one_spin = (not TI) and estnoise is noisest and wt is not None
    if one_spin:
        nsp = tuple(1 for _ in range(x.dim()))
    if ((TI or one_spin) and isinstance(wt, OrthoFilter) and (x.dim() == 1 or (x.dim() == 2 and len(nsp) == 2)) and isinstance(dnt.th, THType)
            and dnt.th.code is not None and 0 <= dnt.th.code <= 3):
        # the translation-invariant branch for orthogonal filters, vectors and square matrices: one device-resident batch
        # (wl_denoise_ti_filter) -- all spins transformed / thresholded / inverted together, the noise estimate consumed on
        # the device, nothing allocated per spin, no host synchronisation.  Same arithmetic in the same order as the loop below.
        sig = -1.0 if estnoise is noisest else float(estnoise(x, wt))
        if estnoise is not noisest and not (sig >= 0 and sig * float(dnt.t) >= 0):
            raise AssertionError("t >= 0")               # (@assert t >= 0 in threshold!, threshold_main.jl:24, t = sigma * dnt.t: NaN
            #                                              fails it, +Inf passes -- everything is thresholded -- as in the reference)
        y = similar(x)
        h, st = _context(x.device)
        q = np.ascontiguousarray(wt.qmf, dtype=np.float64)
        # vectors: prod(nspin) spins shifted by 0 .. pns-1 whatever nspin's length (denoising.jl:38-42)
        nsl = [int(np.prod(nsp))] if x.dim() == 1 else list(nsp)
        nsv = (C.c_int64 * 3)(*(nsl + [1] * (3 - len(nsl))))
        rc = _lib.load().wl_denoise_ti_filter(h, _dtype_code(x), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), x.dim(), _dims(x),
                                              q.ctypes.data_as(C.POINTER(C.c_double)), len(q), int(L), dnt.th.code, float(dnt.t), nsv,
                                              sig, st)
        if _lib.STATUS.get(rc) != "WL_ENOMEM":
            _check(rc, h)
            return y
        # (no room for the batch's workspace -- about ws + 2 N prod(nspin) elements, 5.5 N for one spin: the reference's own
        #  sequence below runs the same device kernels one spin at a time and needs only the transform's two buffers)
    if (TI and isinstance(wt, GLS) and (x.dim() == 1 or (x.dim() == 2 and len(nsp) == 2)) and isinstance(dnt.th, THType)
            and dnt.th.code is not None and 0 <= dnt.th.code <= 3):
        # the same device-resident batch for a lifting scheme (wl_denoise_ti_lifting, round 4): shifted signals as one
        # batched-lines transform, shifted images one 2-D lifting transform per plane; sigma never leaves the device
        sig = -1.0 if estnoise is noisest else float(estnoise(x, wt))
        if estnoise is not noisest and not (sig >= 0 and sig * float(dnt.t) >= 0):
            raise AssertionError("t >= 0")
        y = similar(x)
        h, st = _context(x.device)
        iu, nc, sh, cf = wt.flatten()
        nsl = [int(np.prod(nsp))] if x.dim() == 1 else list(nsp)
        nsv = (C.c_int64 * 3)(*(nsl + [1] * (3 - len(nsl))))
        rc = _lib.load().wl_denoise_ti_lifting(h, _dtype_code(x), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), x.dim(), _dims(x),
                                               len(iu), _i32p(iu), _i32p(nc), _i32p(sh), _f64p(cf), wt.norm1, wt.norm2,
                                               int(L), dnt.th.code, float(dnt.t), nsv, sig, st)
        if _lib.STATUS.get(rc) != "WL_ENOMEM":
            _check(rc, h)
            return y
        # (as above: the per-spin loop below needs far less memory than the batch)
    sigma = estnoise(x, wt)
    t = sigma * dnt.t
    if TI:
        if wt is None:
            raise RuntimeError("TI not supported with wt=nothing")
        pns = int(np.prod(nsp))
        y = similar(x)
        y.zero_()
        xt = similar(x)
        for i in range(1, pns + 1):
            shift = [i - 1] if x.dim() == 1 else nspin2circ(nsp, i)      # (denoising.jl:40-42 / :52-53)
            z = circshift(x, shift)
            dwt_oop_(xt, z, wt, L)
            threshold_(xt, dnt.th, t)
            idwt_oop_(z, xt, wt, L)
            _arrayadd_(y, circshift(z, [-s for s in shift]))
        return _rmul_(y, 1 / pns)
    if wt is None:
        y = similar(x)
        y.copy_(x)
        return threshold_(y, dnt.th, t)
    y = dwt(x, wt, L)
    threshold_(y, dnt.th, t)
    if isinstance(wt, GLS):
        return idwt_(y, wt, L)
    return idwt(y, wt, L)
