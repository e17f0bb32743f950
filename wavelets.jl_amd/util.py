"""Util -- index/size helpers on the transform path (host-side mirror of src/Util)."""
from __future__ import annotations

import numpy as np


def sufficientpoweroftwo(n_or_shape, L: int) -> bool:
    """util_main.jl:21-27"""
    if isinstance(n_or_shape, (int, np.integer)):
        return int(n_or_shape) % (2 ** int(L)) == 0
    return all(int(n) % (2 ** int(L)) == 0 for n in n_or_shape)


def maxtransformlevels(n_or_x) -> int:
    """non_dyadic.jl:14-23 -- for an array: the minimum over its dimensions."""
    if hasattr(n_or_x, "shape"):
        return min(maxtransformlevels(int(s)) for s in n_or_x.shape)
    n = int(n_or_x)
    if n <= 1:
        return 0
    tl = 0
    while sufficientpoweroftwo(n, tl):
        tl += 1
    return tl - 1


def _size1(x):
    return int(x.shape[0]) if hasattr(x, "shape") else int(x)


def detailindex(x, l: int, i: int) -> int:
    """non_dyadic.jl:5 (1-based, as in the reference)"""
    return int(round(_size1(x) / 2 ** l + i))


def detailrange(x, l: int) -> range:
    """non_dyadic.jl:7 (1-based inclusive range)"""
    n = _size1(x)
    return range(int(round(n / 2 ** l + 1)), int(round(n / 2 ** (l - 1))) + 1)


def detailn(x, l: int) -> int:
    """non_dyadic.jl:11"""
    return int(round(_size1(x) / 2 ** l))


def ndyadicscales(x) -> int:
    """dyadic.jl:11"""
    return int(round(np.log2(_size1(x))))


def iscube(x) -> bool:
    """util_main.jl:4-9"""
    return all(s == x.shape[0] for s in x.shape)


def isdyadic(x) -> bool:
    if hasattr(x, "shape"):
        return all(isdyadic(int(s)) for s in x.shape)
    n = int(x)
    return n == 2 ** ndyadicscales(n)


def maketree(n_or_x, L: int = None, s: str = "full") -> np.ndarray:
    """util_main.jl:316-344 -- BitVector as a uint8 array (1 byte per node)."""
    n = len(n_or_x) if hasattr(n_or_x, "__len__") else int(n_or_x)
    ns = maxtransformlevels(n)
    if L is None:
        L = ns
    nb = 2 ** ns - 1
    assert 0 <= L <= ns
    b = np.zeros(nb, dtype=np.uint8)
    if s == "full":
        b[: 2 ** L - 1] = 1
    elif s == "dwt":
        for i in range(1, L + 1):
            b[2 ** (i - 1) - 1] = 1
    else:
        raise ValueError("uknown symbol")
    return b


def isvalidtree(x, b) -> bool:
    """util_main.jl:301-314"""
    n = len(x) if hasattr(x, "__len__") else int(x)
    ns = maxtransformlevels(n)
    nb = len(b)
    if nb != 2 ** ns - 1:
        return False
    for i in range(1, 2 ** (ns - 1)):
        if not b[i - 1] and (b[2 * i - 1] or b[2 * i]):
            return False
    return True


# ---- dyadic indexing (dyadic.jl:3-20; 1-based indices / ranges as in the reference) ---------------------
def dyadicdetailindex(j: int, i: int) -> int:
    return 2 ** j + i


def dyadicdetailrange(j: int) -> range:
    return range(2 ** j + 1, 2 ** (j + 1) + 1)


def dyadicscalingrange(j: int) -> range:
    return range(1, 2 ** j + 1)


def dyadicdetailn(j: int) -> int:
    return 2 ** j


def maxdyadiclevel(x) -> int:
    return ndyadicscales(x) - 1


def tl2dyadiclevel(x, L: int) -> int:
    n = len(x) if hasattr(x, "__len__") else int(x)
    return ndyadicscales(n) - int(L)


dyadiclevel2tl = tl2dyadiclevel


# ---- small vector helpers (util_main.jl:30-81); numpy arrays or device tensors ---------------------------
def mirror(f):
    """f .* (-1).^(0:length(f)-1)"""
    f = np.asarray(f)
    return f * (-1.0) ** np.arange(len(f))


def upsample(x, sw: int = 0):
    """zero-stuffing: y[2i + sw] = x[i] (0-based i)"""
    assert sw in (0, 1)
    if hasattr(x, "new_zeros"):            # torch tensor (host or device)
        y = x.new_zeros(2 * x.shape[0])
    else:
        x = np.asarray(x)
        y = np.zeros(2 * len(x), dtype=x.dtype)
    y[1 - 1 + sw::2] = x
    return y


def downsample(x, sw: int = 0):
    """y[i] = x[2i + sw] (0-based i)"""
    assert sw in (0, 1)
    assert len(x) % 2 == 0
    return x[sw::2].clone() if hasattr(x, "clone") else np.asarray(x)[sw::2].copy()


def wcount(x, t=0, level: int = -1) -> int:
    """number of coefficients with abs(x) >= t, for vectors excluding the levels below `level` (level -1 is x[1])"""
    assert level >= -1
    if getattr(x, "ndim", 1) == 1 and level >= 0:
        x = x[2 ** level:]
    if hasattr(x, "abs") and hasattr(x, "device"):
        return int((x.abs() >= t).sum().item())
    return int((np.abs(np.asarray(x)) >= t).sum())


def testfunction(n: int, ft: str) -> np.ndarray:
    """The Donoho-Johnstone test signals sampled at t = 0, 1/n, ..., (n-1)/n (util_main.jl:378-420), Float64."""
    assert n >= 1
    t = np.arange(n, dtype=np.float64) / n
    if ft in ("Blocks", "Bumps"):
        tj = np.array([0.1, 0.13, 0.15, 0.23, 0.25, 0.4, 0.44, 0.65, 0.76, 0.78, 0.81])
        f = np.zeros(n)
        if ft == "Blocks":
            hj = [4, -5, 3, -4, 5, -4.2, 2.1, 4.3, -3.1, 2.1, -4.2]
            for h, tk in zip(hj, tj):                      # accumulated term by term, in the reference's order
                f = f + h * (1 + np.sign(t - tk)) / 2
        else:
            hj = [4, 5, 3, 4, 5, 4.2, 2.1, 4.3, 3.1, 5.1, 4.2]
            wj = [0.005, 0.005, 0.006, 0.01, 0.01, 0.03, 0.01, 0.01, 0.005, 0.008, 0.005]
            for h, tk, wk in zip(hj, tj, wj):
                f = f + h / (1 + np.abs((t - tk) / wk)) ** 4
        return f
    if ft == "HeaviSine":
        return 4 * np.sin(4 * np.pi * t) - np.sign(t - 0.3) - np.sign(0.72 - t)
    if ft == "Doppler":
        return np.sqrt(t * (1 - t)) * np.sin(2 * np.pi * 1.05 / (t + 0.05))
    raise ValueError("unknown test function")          # ArgumentError in the reference
