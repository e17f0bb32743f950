"""Maximal-overlap DWT -- host-side mirror of src/Transforms/transforms_maximal_overlap.jl
(modwt :47-63, imodwt :99-107) and maxmodwttransformlevels (src/Util/non_dyadic.jl:24-25).

`modwt(x, wt, L)` returns the n x (L+1) coefficient matrix as a column-major device tensor (each
level's coefficients are one contiguous column; the scaling coefficients are the last column),
`imodwt(xw, wt)` inverts it.  The compute is libwavelets_mi355x.so (wl_modwt / wl_imodwt); there is
no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .transforms import (ArgumentError, DimensionMismatch, HIPError, _check, _context, _dtype_code, _f64p, _prep_in,
                         julia_layout)
from .wt import OrthoFilter


def maxmodwttransformlevels(x) -> int:
    """floor(Int, log2(length(x))) -- length of an array, or the integer itself"""
    n = int(x.numel()) if isinstance(x, torch.Tensor) else (int(np.asarray(x).size) if hasattr(x, "__len__") else int(x))
    if n < 1:
        raise ArgumentError("maxmodwttransformlevels of an empty array (DomainError in the reference)")
    return int(_lib.load().wl_maxmodwttransformlevels(n))


def modwt(x, wt: OrthoFilter, L: Optional[int] = None) -> torch.Tensor:
    """modwt(x::AbstractVector, wt::OrthoFilter, L=maxmodwttransformlevels(x)) -> n x (L+1)"""
    if not isinstance(wt, OrthoFilter):
        raise TypeError("modwt is defined for OrthoFilter wavelets only (MethodError in the reference)")
    x = _prep_in(x)
    if x.dim() != 1:
        raise TypeError("modwt expects a vector (MethodError in the reference)")
    n = int(x.numel())
    if n < 1:
        raise ArgumentError("modwt of an empty vector")
    L = maxmodwttransformlevels(n) if L is None else int(L)
    if L > maxmodwttransformlevels(n):
        raise ArgumentError("Too many transform levels (length(x) < 2^L)")
    if L < 1:
        raise ArgumentError("L must be >= 1")
    out = torch.empty((L + 1, n), dtype=x.dtype, device=x.device).t()       # column-major n x (L+1)
    h, st = _context(x.device)
    q = np.ascontiguousarray(wt.qmf, dtype=np.float64)
    rc = _lib.load().wl_modwt(h, _dtype_code(x), C.c_void_p(out.data_ptr()), n, C.c_void_p(x.data_ptr()), n,
                              _f64p(q), len(q), L, st)
    _check(rc, h)
    return out


def imodwt(xw, wt: OrthoFilter) -> torch.Tensor:
    """imodwt(xw::Matrix, wt::OrthoFilter): inverse of modwt(x, wt, size(xw, 2) - 1)"""
    if not isinstance(wt, OrthoFilter):
        raise TypeError("imodwt is defined for OrthoFilter wavelets only (MethodError in the reference)")
    if not isinstance(xw, torch.Tensor) or xw.device.type != "cuda":
        raise HIPError("expected a torch tensor resident on an MI355X device; there is no CPU path")
    if xw.dim() != 2:
        raise TypeError("imodwt expects a matrix (MethodError in the reference)")
    xw = julia_layout(xw)
    n, ncols = int(xw.shape[0]), int(xw.shape[1])
    if n < 1 or ncols < 1:
        raise DimensionMismatch("empty coefficient matrix")
    x = torch.empty(n, dtype=xw.dtype, device=xw.device)
    h, st = _context(xw.device)
    q = np.ascontiguousarray(wt.qmf, dtype=np.float64)
    rc = _lib.load().wl_imodwt(h, _dtype_code(xw), C.c_void_p(x.data_ptr()), C.c_void_p(xw.data_ptr()), n, n, ncols,
                               _f64p(q), len(q), st)
    _check(rc, h)
    return x
