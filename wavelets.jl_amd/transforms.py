"""Transforms -- host-side mirror of the reference's public transform API
(src/Transforms/transforms_main.jl:105-176): dwt / idwt / dwt! / idwt! / wpt / iwpt /
wpt! / iwpt!, plus dwtc / idwtc (named at :179-181, defined by this build as the 1-D
transform of every column).  Julia's `f!` is spelled `f_` here.

Every function validates its arguments the way the reference's `_dwt!` does
(transforms_filter.jl:24-38, transforms_lifting.jl:33-43,131-143), then calls the C ABI of
libwavelets_mi355x.so (include/wavelets_mi355x.h) on the tensor's device and current
stream.  Arrays are torch tensors resident in HBM, in *Julia layout* (column-major: for a
2-D tensor of shape (m, n) the strides are (1, m)), so element [i, j] here is element
[i+1, j+1] of the Julia array.  `to_device` builds such tensors from numpy arrays.

There is no CPU implementation behind these functions: a CPU tensor / numpy array raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from . import util as Util
from .wt import GLS, OrthoFilter


class DimensionMismatch(ValueError):
    """Julia's DimensionMismatch"""


class ArgumentError(ValueError):
    """Julia's ArgumentError"""


class HIPError(RuntimeError):
    pass


_STATUS_EXC = {
    -1: ArgumentError, -2: ArgumentError, -3: ArgumentError, -4: DimensionMismatch, -5: ArgumentError,
    -6: ArgumentError, -7: ArgumentError, -8: TypeError, -9: ArgumentError, -10: ArgumentError,
    -11: MemoryError, -12: HIPError, -13: HIPError,
}


def _check(rc: int, ctx=None):
    if rc == 0:
        return
    msg = _lib.strerror(rc)
    if rc == -12 and ctx is not None:
        msg += f" (hipError_t {_lib.load().wl_last_hip_error(ctx)})"
    raise _STATUS_EXC.get(rc, RuntimeError)(f"{_lib.STATUS.get(rc, rc)}: {msg}")


# ---- contexts: one per (device, stream) ------------------------------------------------------
_CTX = {}
_FORCE_PATH = 0      # tests flip this to 1 to run the generic kernels only
_OPTIONS = {}        # wl_ctx_set_option pairs applied to every context (existing and future)


def set_option(key: str, value: int):
    """Set a tuning / test switch (wl_ctx_set_option) on every context of this process.  No option changes a
    result; the library reads no environment variables."""
    _OPTIONS[str(key)] = int(value)
    for h in _CTX.values():
        _check(_lib.load().wl_ctx_set_option(h, str(key).encode(), int(value)))


def clear_options():
    _OPTIONS.clear()
    for h in _CTX.values():
        _check(_lib.load().wl_ctx_clear_options(h))


class options:
    """with W.options(WL_FUSE2_MIN=0): ...   -- options set for the block, previous table restored afterwards."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.saved = dict(_OPTIONS)
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        saved = self.saved
        clear_options()
        for k, v in saved.items():
            set_option(k, v)
        return False


def set_kernel_path(path: int):
    """0 = auto (fast paths), 1 = generic kernels only.  Results are bit-identical."""
    global _FORCE_PATH
    _FORCE_PATH = int(path)
    for h in _CTX.values():
        _check(_lib.load().wl_ctx_set_path(h, _FORCE_PATH))


def set_arithmetic(mode: str):
    """"exact" (default): every product and sum rounded separately, results bit-identical to the reference's CPU path.
    "fused": the opt-in build of the same kernels with FMA contraction allowed (libwavelets_mi355x_fma.so) -- results agree
    with the reference to SURVEY.md 8(c)'s tolerances (f32 relative L2 <= 1e-6*sqrt(L), f64 <= 1e-13*sqrt(L)), not bit for
    bit.  Process-wide; the contexts (and workspaces) of the library being left are destroyed first."""
    if mode == _lib.arithmetic():
        return
    if mode not in _lib.LIB_PATHS:
        raise ValueError(f"arithmetic mode must be one of {sorted(_lib.LIB_PATHS)}, got {mode!r}")
    # resolve and bind the target library FIRST: a missing or stale libwavelets_mi355x_fma.so must leave the process in the mode
    # it was in, with its contexts alive (round-5 review)
    _lib.load(mode)
    if _CTX:
        destroy_contexts()
    _lib._select(mode)


def get_arithmetic() -> str:
    return _lib.arithmetic()


class arithmetic:
    """with W.arithmetic("fused"): ...   -- mode for the block, previous mode restored afterwards."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.saved = _lib.arithmetic()
        set_arithmetic(self.mode)
        return self

    def __exit__(self, *exc):
        set_arithmetic(self.saved)
        return False


def _context(device: torch.device):
    if device.type != "cuda":
        raise HIPError("wavelets_jl_amd runs on MI355X (gfx950) HIP devices only; there is no CPU path")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(idx).cuda_stream
    key = (idx, stream)
    h = _CTX.get(key)
    if h is None:
        lib = _lib.load()
        out = C.c_void_p()
        _check(lib.wl_ctx_create(idx, C.byref(out)))
        h = out
        _check(lib.wl_ctx_set_path(h, _FORCE_PATH))
        for k, v in _OPTIONS.items():
            _check(lib.wl_ctx_set_option(h, k.encode(), v))
        _CTX[key] = h
    return h, C.c_void_p(stream)


def destroy_contexts():
    """Destroy every library context this process created (one per device and stream, each owning a grow-only device
    workspace) -- e.g. after a burst of work on temporary streams.  New contexts are created on demand."""
    lib = _lib.load()
    torch.cuda.synchronize()
    for h in list(_CTX.values()):
        lib.wl_ctx_destroy(h)
    _CTX.clear()


def last_kernel(device=None) -> str:
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    h, _ = _context(dev)
    return _lib.load().wl_last_kernel(h).decode()


def workspace_held(device=None) -> int:
    """bytes of device workspace currently held by the context of the current stream (grow-only)"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    h, _ = _context(dev)
    return int(_lib.load().wl_ctx_workspace_held(h))


def reserve_workspace(x: torch.Tensor, L: int = 0, full: bool = False):
    """Pre-grow the context's device workspace for transforms of x's shape.  Default: what the fast filter-bank paths of
    dwt / idwt / dwtc hold (two approximation buffers of N / 2^ndims elements) -- after this, those calls never allocate.
    full=True reserves the upper bound of EVERY entry point (lifting, long / odd filters, 3-D, generic kernels, wpt: about
    4.5 N elements), so that no transform of this shape allocates or synchronises whatever path it takes."""
    lib = _lib.load()
    h, _ = _context(x.device)
    dims = (C.c_int64 * 3)(*([int(s) for s in x.shape] + [1] * (3 - x.dim())))
    fn = lib.wl_workspace_bytes_full if full else lib.wl_workspace_bytes
    nbytes = fn(_dtype_code(x), x.dim(), dims, int(L))
    _check(lib.wl_ctx_reserve(h, nbytes), h)


# ---- Julia-layout tensors ---------------------------------------------------------------------
def _jl_strides(shape):
    st, acc = [], 1
    for s in shape:
        st.append(acc)
        acc *= int(s)
    return tuple(st)


def is_julia_layout(x: torch.Tensor) -> bool:
    return tuple(x.stride()) == _jl_strides(x.shape) or x.numel() <= 1


def similar(x: torch.Tensor, dtype=None) -> torch.Tensor:
    """similar(x): an uninitialised column-major tensor of the same shape on the same device."""
    shape = tuple(x.shape)
    base = torch.empty(tuple(reversed(shape)), dtype=dtype or x.dtype, device=x.device)
    return base.permute(*reversed(range(len(shape)))) if len(shape) > 1 else base


def julia_layout(x: torch.Tensor) -> torch.Tensor:
    """Return x itself when it already is column-major dense, else a column-major copy."""
    if is_julia_layout(x):
        return x
    y = similar(x)
    y.copy_(x)
    return y


def to_device(a, device="cuda", dtype=None) -> torch.Tensor:
    """numpy array (any memory order) -> column-major device tensor with the same logical indices."""
    a = np.asarray(a)
    if dtype is not None:
        a = a.astype(dtype)
    t = torch.from_numpy(np.ascontiguousarray(a.T))          # row-major of the transpose == column-major of a
    t = t.to(device)
    return t.permute(*reversed(range(a.ndim))) if a.ndim > 1 else t


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy()


def _dtype_code(x: torch.Tensor) -> int:
    if x.dtype == torch.float32:
        return _lib.WL_F32
    if x.dtype == torch.float64:
        return _lib.WL_F64
    raise TypeError(f"element type {x.dtype} is not supported (Float32/Float64 only; Complex is out of scope)")


def _prep_in(x) -> torch.Tensor:
    if not isinstance(x, torch.Tensor):
        raise TypeError("expected a torch tensor resident on an MI355X device (use to_device(array)); "
                        "there is no CPU path")
    if x.device.type != "cuda":
        raise HIPError("tensor is not on a HIP device; there is no CPU path")
    if x.dim() < 1 or x.dim() > 3:
        raise DimensionMismatch("only 1-D, 2-D and 3-D arrays are supported")
    if not x.dtype.is_floating_point and not x.dtype.is_complex:
        x = x.to(torch.float64)        # Int -> Float (transforms_main.jl:188-190)
    _dtype_code(x)
    return julia_layout(x)


def _dims(x: torch.Tensor):
    return (C.c_int64 * 3)(*([int(s) for s in x.shape] + [1] * (3 - x.dim())))


def _f64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _default_L(x, L):
    return Util.maxtransformlevels(x) if L is None else int(L)


# ---- core calls ----------------------------------------------------------------------------------
def _check_one(y: torch.Tensor, vector_only: bool = False):
    """What every caller-supplied array must satisfy before its data_ptr() reaches the C ABI."""
    if not isinstance(y, torch.Tensor):
        raise TypeError("expected a torch tensor resident on the GPU")
    if y.device.type != "cuda":
        raise HIPError("wavelets_jl_amd runs on MI355X (gfx950) HIP devices only; there is no CPU path")
    if y.dtype not in (torch.float32, torch.float64):
        raise TypeError("element type must be Float32 or Float64")
    if vector_only and y.dim() != 1:
        raise TypeError("wpt is defined for vectors only (WPTArray = AbstractVector)")
    if not is_julia_layout(y):
        raise ArgumentError("arrays must be dense column-major (Julia layout); see to_device/similar")


def _check_pair(y: torch.Tensor, x: torch.Tensor, vector_only: bool = False):
    """(y, x) handed to an out-of-place entry point: same shape (DimensionMismatch, transforms_filter.jl:25-26), same
    element type, same device, both dense column-major -- the library trusts the extents it is given."""
    _check_one(y, vector_only)
    _check_one(x, vector_only)
    if tuple(x.shape) != tuple(y.shape):
        raise DimensionMismatch("in and out array size must match")
    if x.dtype != y.dtype:
        raise TypeError("x and y must have the same element type")
    if x.device != y.device:
        raise HIPError("x and y are on different devices")


def _filter_call(y: torch.Tensor, x: torch.Tensor, filt: OrthoFilter, L: int, fw: bool):
    _check_pair(y, x)
    lib = _lib.load()
    h, st = _context(x.device)
    q = np.ascontiguousarray(filt.qmf, dtype=np.float64)
    rc = lib.wl_dwt_filter(h, _dtype_code(x), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()),
                           x.dim(), _dims(x), _f64p(q), len(q), int(L), 1 if fw else 0, st)
    _check(rc, h)
    return y


def _lifting_call(y: torch.Tensor, x: Optional[torch.Tensor], sch: GLS, L: int, fw: bool):
    if x is None:
        _check_one(y)
    else:
        _check_pair(y, x)
    lib = _lib.load()
    h, st = _context(y.device)
    iu, nc, sh, cf = sch.flatten()
    if x is None:
        rc = lib.wl_dwt_lifting(h, _dtype_code(y), C.c_void_p(y.data_ptr()), y.dim(), _dims(y),
                                len(iu), _i32p(iu), _i32p(nc), _i32p(sh), _f64p(cf),
                                sch.norm1, sch.norm2, int(L), 1 if fw else 0, st)
    else:
        rc = lib.wl_dwt_lifting_oop(h, _dtype_code(y), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()),
                                    y.dim(), _dims(y), len(iu), _i32p(iu), _i32p(nc), _i32p(sh), _f64p(cf),
                                    sch.norm1, sch.norm2, int(L), 1 if fw else 0, st)
    _check(rc, h)
    return y


def _xwt(x, wt, L, fw):
    x = _prep_in(x)
    L = _default_L(x, L)
    if x.numel() == 0:                     # empty array: maxtransformlevels == 0, the transform is a copy
        if L != 0 and L is not None and int(L) > 0:
            raise ArgumentError("size must have a sufficient power of 2 factor")
        return similar(x)
    if isinstance(wt, OrthoFilter):
        return _filter_call(similar(x), x, wt, L, fw)                 # transforms_main.jl:109-113
    if isinstance(wt, GLS):
        return _lifting_call(similar(x), x, wt, L, fw)               # :119-124 (copy fused away)
    raise TypeError("wt must be an OrthoFilter or a GLS (MethodError in the reference)")


def dwt(x, wt, L: Optional[int] = None) -> torch.Tensor:
    """dwt(x, wt[, L=maxtransformlevels(x)])"""
    return _xwt(x, wt, L, True)


def idwt(x, wt, L: Optional[int] = None) -> torch.Tensor:
    """idwt(x, wt[, L=maxtransformlevels(x)])"""
    return _xwt(x, wt, L, False)


def _xwt_inplace(args, fw):
    # dwt!(y, x, filter[, L])  |  dwt!(y, scheme[, L])      (transforms_main.jl:114-128)
    if len(args) >= 3 and isinstance(args[2], OrthoFilter):
        y, x, filt = args[0], args[1], args[2]
        L = _default_L(x, args[3] if len(args) > 3 else None)
        if y is x or (isinstance(y, torch.Tensor) and isinstance(x, torch.Tensor) and y.data_ptr() == x.data_ptr()
                      and tuple(y.shape) == tuple(x.shape)):
            raise ArgumentError("in array is out array")
        for t in (y, x):
            if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
                raise HIPError("arrays must be torch tensors on a HIP device; there is no CPU path")
        return _filter_call(y, x, filt, L, fw)
    if len(args) >= 2 and isinstance(args[1], GLS):
        y, sch = args[0], args[1]
        if not isinstance(y, torch.Tensor) or y.device.type != "cuda":
            raise HIPError("array must be a torch tensor on a HIP device; there is no CPU path")
        L = _default_L(y, args[2] if len(args) > 2 else None)
        return _lifting_call(y, None, sch, L, fw)
    raise TypeError("usage: dwt_(y, x, filter[, L]) or dwt_(y, scheme[, L])")


def dwt_(*args) -> torch.Tensor:
    """dwt!(y, x, wt::OrthoFilter[, L]) / dwt!(y, wt::GLS[, L])"""
    return _xwt_inplace(args, True)


def idwt_(*args) -> torch.Tensor:
    """idwt!(y, x, wt::OrthoFilter[, L]) / idwt!(y, wt::GLS[, L])"""
    return _xwt_inplace(args, False)


def dwt_oop_(y, x, wt, L: Optional[int] = None) -> torch.Tensor:
    """dwt_oop!(y, x, wt[, L]) (non-exported in the reference, transforms_main.jl:193-207): out of place
    into a caller-provided y for filters AND lifting schemes (the reference copies x into y and runs in
    place; here the copy is fused away)."""
    L = _default_L(x, L)
    if isinstance(wt, OrthoFilter):
        return _xwt_inplace((y, x, wt, L), True)
    if isinstance(wt, GLS):
        if tuple(x.shape) != tuple(y.shape):
            raise DimensionMismatch("in and out array size must match")
        return _lifting_call(y, x, wt, L, True)
    raise TypeError("wt must be an OrthoFilter or a GLS")


def idwt_oop_(y, x, wt, L: Optional[int] = None) -> torch.Tensor:
    L = _default_L(x, L)
    if isinstance(wt, OrthoFilter):
        return _xwt_inplace((y, x, wt, L), False)
    if isinstance(wt, GLS):
        if tuple(x.shape) != tuple(y.shape):
            raise DimensionMismatch("in and out array size must match")
        return _lifting_call(y, x, wt, L, False)
    raise TypeError("wt must be an OrthoFilter or a GLS")


# ---- batched column-wise ----------------------------------------------------------------------
def _xwtc(x, wt, L, fw, y=None):
    x = _prep_in(x)
    if x.dim() != 2:
        raise DimensionMismatch("dwtc expects a len x nsignals matrix")
    length, nsig = int(x.shape[0]), int(x.shape[1])
    L = Util.maxtransformlevels(length) if L is None else int(L)
    lib = _lib.load()
    h, st = _context(x.device)
    if y is not None and (tuple(y.shape) != tuple(x.shape) or y.dtype != x.dtype or not is_julia_layout(y)):
        raise DimensionMismatch("in and out array size must match")
    if isinstance(wt, OrthoFilter):
        if y is None:
            y = similar(x)
        q = np.ascontiguousarray(wt.qmf, dtype=np.float64)
        rc = lib.wl_dwtc_filter(h, _dtype_code(x), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()),
                                length, nsig, length, _f64p(q), len(q), L, 1 if fw else 0, st)
        _check(rc, h)
        return y
    if isinstance(wt, GLS):
        if y is None:
            y = similar(x)
        iu, nc, sh, cf = wt.flatten()
        # out of place straight from x (no copy, no staging of the in-place first level); y is x: the in-place transform
        rc = lib.wl_dwtc_lifting_oop(h, _dtype_code(y), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), length, nsig, length,
                                     len(iu), _i32p(iu), _i32p(nc), _i32p(sh), _f64p(cf), wt.norm1, wt.norm2,
                                     L, 1 if fw else 0, st)
        _check(rc, h)
        return y
    raise TypeError("wt must be an OrthoFilter or a GLS")


def dwtc(x, wt, L: Optional[int] = None) -> torch.Tensor:
    """Column-wise dwt: the 1-D dwt of every column of a len x nsignals matrix."""
    return _xwtc(x, wt, L, True)


def idwtc(x, wt, L: Optional[int] = None) -> torch.Tensor:
    return _xwtc(x, wt, L, False)


def dwtc_(y, x, wt, L: Optional[int] = None) -> torch.Tensor:
    """column-wise dwt into a caller-provided y (no allocation)"""
    return _xwtc(x, wt, L, True, y)


def idwtc_(y, x, wt, L: Optional[int] = None) -> torch.Tensor:
    return _xwtc(x, wt, L, False, y)


# ---- a batch of independent images --------------------------------------------------------------
def _xwt_batch(x, wt, L, fw, y=None):
    """x: n0 x n1 x B (column-major: image i = x[:, :, i]); every image gets its own 2-D transform, all in one chain of
    launches (wl_dwt_filter_batch).  The reference has no batched form: this equals `stack(dwt(x[:, :, i], wt, L) for i)`."""
    x = _prep_in(x)
    if x.dim() != 3:
        raise TypeError("dwt_batch expects an n0 x n1 x B array")
    if not isinstance(wt, OrthoFilter):
        raise TypeError("dwt_batch is defined for orthogonal filters")
    n0, n1, nb = (int(v) for v in x.shape)
    L = min(Util.maxtransformlevels(n0), Util.maxtransformlevels(n1)) if L is None else int(L)
    y = similar(x) if y is None else y
    _check_pair(y, x)                        # (a caller-supplied y: same shape / type / device, dense column-major)
    lib = _lib.load()
    h, st = _context(x.device)
    q = np.ascontiguousarray(wt.qmf, dtype=np.float64)
    dims = (C.c_int64 * 2)(n0, n1)
    rc = lib.wl_dwt_filter_batch(h, _dtype_code(x), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), dims, nb, n0 * n1,
                                 _f64p(q), len(q), L, 1 if fw else 0, st)
    _check(rc, h)
    return y


def dwt_batch(x, wt, L: Optional[int] = None, y=None) -> torch.Tensor:
    return _xwt_batch(x, wt, L, True, y)


def idwt_batch(x, wt, L: Optional[int] = None, y=None) -> torch.Tensor:
    return _xwt_batch(x, wt, L, False, y)


# ---- wavelet packet transforms -------------------------------------------------------------------
class _FullTree(int):
    """wpt(x, wt, L::Integer): the full tree of depth L -- never materialised (a tree has n - 1 nodes; the library's
    wl_wpt_*_full entry points take the depth)."""


def _tree_arg(n, tree_or_L):
    if tree_or_L is None:
        return _FullTree(Util.maxtransformlevels(n))
    if isinstance(tree_or_L, (int, np.integer)):
        L = int(tree_or_L)
        if not 0 <= L <= Util.maxtransformlevels(n):
            raise AssertionError("0 <= L <= maxtransformlevels(n)")       # maketree's @assert (util_main.jl:316-344)
        return _FullTree(L)
    t = np.asarray(tree_or_L)
    if t.dtype == np.bool_:
        t = t.view(np.uint8)                 # BitVector -> one byte per node, no copy
    return np.ascontiguousarray(t, dtype=np.uint8)


def _wpt_filter_call(y, x, filt, tree, fw):
    _check_pair(y, x, vector_only=True)
    lib = _lib.load()
    h, st = _context(x.device)
    q = np.ascontiguousarray(filt.qmf, dtype=np.float64)
    if isinstance(tree, _FullTree):
        rc = lib.wl_wpt_filter_full(h, _dtype_code(x), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), x.numel(),
                                    _f64p(q), len(q), int(tree), 1 if fw else 0, st)
        _check(rc, h)
        return y
    rc = lib.wl_wpt_filter(h, _dtype_code(x), C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), x.numel(),
                           _f64p(q), len(q), tree.ctypes.data_as(C.POINTER(C.c_uint8)), len(tree),
                           1 if fw else 0, st)
    _check(rc, h)
    return y


def _wpt_lifting_call(y, sch, tree, fw):
    _check_one(y, vector_only=True)
    lib = _lib.load()
    h, st = _context(y.device)
    iu, nc, sh, cf = sch.flatten()
    if isinstance(tree, _FullTree):
        rc = lib.wl_wpt_lifting_full(h, _dtype_code(y), C.c_void_p(y.data_ptr()), y.numel(), len(iu), _i32p(iu), _i32p(nc),
                                     _i32p(sh), _f64p(cf), sch.norm1, sch.norm2, int(tree), 1 if fw else 0, st)
        _check(rc, h)
        return y
    rc = lib.wl_wpt_lifting(h, _dtype_code(y), C.c_void_p(y.data_ptr()), y.numel(), len(iu), _i32p(iu), _i32p(nc),
                            _i32p(sh), _f64p(cf), sch.norm1, sch.norm2,
                            tree.ctypes.data_as(C.POINTER(C.c_uint8)), len(tree), 1 if fw else 0, st)
    _check(rc, h)
    return y


def _xwpt(x, wt, tree_or_L, fw):
    x = _prep_in(x)
    if x.dim() != 1:
        raise TypeError("wpt is defined for vectors only (WPTArray = AbstractVector)")
    tree = _tree_arg(x.numel(), tree_or_L)
    if isinstance(wt, OrthoFilter):
        return _wpt_filter_call(similar(x), x, wt, tree, fw)
    if isinstance(wt, GLS):
        y = similar(x)
        y.copy_(x)
        return _wpt_lifting_call(y, wt, tree, fw)
    raise TypeError("wt must be an OrthoFilter or a GLS")


def wpt(x, wt, tree_or_L=None) -> torch.Tensor:
    """wpt(x, wt[, L | tree])"""
    return _xwpt(x, wt, tree_or_L, True)


def iwpt(x, wt, tree_or_L=None) -> torch.Tensor:
    return _xwpt(x, wt, tree_or_L, False)


def _xwpt_inplace(args, fw):
    if len(args) >= 3 and isinstance(args[2], OrthoFilter):
        y, x, filt = args[0], args[1], args[2]
        if y is x or y.data_ptr() == x.data_ptr():
            raise ArgumentError("in array is out array")
        tree = _tree_arg(x.numel(), args[3] if len(args) > 3 else None)
        return _wpt_filter_call(y, x, filt, tree, fw)
    if len(args) >= 2 and isinstance(args[1], GLS):
        y, sch = args[0], args[1]
        tree = _tree_arg(y.numel(), args[2] if len(args) > 2 else None)
        return _wpt_lifting_call(y, sch, tree, fw)
    raise TypeError("usage: wpt_(y, x, filter[, L | tree]) or wpt_(y, scheme[, L | tree])")


def wpt_(*args) -> torch.Tensor:
    """wpt!(y, x, filter[, L | tree]) / wpt!(y, scheme[, L | tree])"""
    return _xwpt_inplace(args, True)


def iwpt_(*args) -> torch.Tensor:
    return _xwpt_inplace(args, False)
