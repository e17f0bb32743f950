"""numpy statement of the closed forms the HIP kernels evaluate (wl_internal.h), with the same
summation order and separate multiply/add roundings.  Used by CPU tests to prove, without a GPU,
that the closed forms are bit-identical to the literal shift-register oracle."""
import numpy as np


def taps(qmf, dtype):
    h = np.asarray(qmf, dtype=np.float64).astype(dtype)
    g = h.copy()
    g[1::2] = g[1::2] * dtype(-1)
    return h, g


def fwd_level(x, h, g):
    """one forward level on a 1-D line: returns (s, d)"""
    n = len(x)
    nx = n // 2
    F = len(h)
    k = np.arange(nx)
    s = h[0] * x[(2 * k) % n]
    for m in range(1, F):
        s = s + h[m] * x[(2 * k + m) % n]
    d = g[F - 1] * x[(2 * k + 1 - (F - 1)) % n]
    for m in range(F - 2, -1, -1):
        d = d + g[m] * x[(2 * k + 1 - m) % n]
    return s.astype(x.dtype), d.astype(x.dtype)


def inv_level(s, d, h, g):
    nx = len(s)
    n = 2 * nx
    F = len(h)
    out = np.empty(n, dtype=s.dtype)
    for par in (0, 1):
        o = np.arange(par, n, 2)
        S = None
        for m in range(F - 1, -1, -1):
            if (par - m) % 2 == 0:
                term = h[m] * s[((o - m) // 2) % nx]
                S = term if S is None else S + term
        D = None
        for m in range(F):
            if (par + m - 1) % 2 == 0:
                term = g[m] * d[((o + m - 1) // 2) % nx]
                D = term if D is None else D + term
        out[o] = S + D
    return out


def dwt1d(x, qmf, L, fw=True):
    h, g = taps(qmf, x.dtype.type)
    n = len(x)
    y = x.copy()
    if fw:
        a = x
        for l in range(1, L + 1):
            s, d = fwd_level(a, h, g)
            y[n >> l: n >> (l - 1)] = d
            a = s
        y[: n >> L] = a
        return y
    a = x[: n >> L]
    for l in range(L, 0, -1):
        a = inv_level(a, x[n >> l: n >> (l - 1)], h, g)
    return a


def lift_step(w, half, is_update, shift, c):
    """one lifting step in place on w = [s ; d] (closed form of lift!, wl_generic.hip)"""
    nc = len(c)
    j = np.arange(half)
    j0 = j - shift
    inb = (j0 >= 0) & (j0 + nc - 1 <= half - 1)
    tgt = w[half:2 * half] if is_update else w[:half]
    opnd = w[:half] if is_update else w[half:2 * half]
    x = tgt.copy()
    acc = c[0] * opnd[j0 % half]
    for k in range(1, nc):
        acc = acc + c[k] * opnd[(j0 + k) % half]
    xin = x + acc
    xb = x.copy()
    for k in range(nc):
        xb = xb + c[k] * opnd[(j0 + k) % half]
    tgt[:] = np.where(inb, xin, xb)
    return w
