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
