"""The Float32 "cut" that the level kernels of the TI denoise use to decide which coefficients a threshold zeroes (ThCut,
wavelets.jl_amd/csrc/wl_dev.h) -- CPU check of the arithmetic claim, independent of the device code (which tests/test_gpu_ext.py
pins against the oracle): the reference compares / subtracts in Float64 (t = sigma * dnt.t is a Float64, x a Float32), and

  hard:            |x| <= t           <=>  |x| <= the largest Float32 <= t
  soft, semisoft:  |x| -  t < 0       <=>  |x| <= the largest Float32 <  t
  Stein:           1 - t^2 / x^2 < 0  <==  |x| <= the largest Float32 <= 0.999 t     (sufficient, not necessary)

for every Float32 x, checked here on the Float32 neighbourhood of t (where a cut could go wrong) and on random values.
Reference: /root/reference/src/Threshold/threshold_main.jl:37-122 (restated in oracle/wl_oracle_ext.c)."""
import numpy as np


def _prev(f):
    return np.nextafter(np.float32(f), np.float32(-np.inf), dtype=np.float32)


def cut(kind, t):
    """mirror of th_make_cut: the largest Float32 |x| that is surely zeroed (-1: no cut)"""
    if not (t > 0):
        return np.float32(0.0) if (kind == "hard" and t == 0) else np.float32(-1.0)
    lim = 0.999 * t if kind == "stein" else t
    f = np.float32(lim)
    strict = kind in ("soft", "semisoft")
    if float(f) > lim or (strict and float(f) == lim):
        f = _prev(f) if f > 0 else np.float32(-1.0)
    if kind != "hard" and np.isinf(f):
        f = np.finfo(np.float32).max
    return f


def zeroed_by_reference(kind, x, t):
    """does threshold!(x, kind, t) produce 0 for the Float32 x (Float64 arithmetic, as Julia's promotion does)"""
    xi = float(x)
    ax = abs(xi)
    if kind == "hard":
        return ax <= t
    if kind == "soft":
        return ax - t < 0
    if kind == "semisoft":
        return xi <= 2 * t and ax - t < 0
    with np.errstate(divide="ignore", invalid="ignore"):
        sh = np.float64(1) - np.float64(t) * np.float64(t) / (np.float64(xi) * np.float64(xi))
    return bool(sh < 0)


def _neighbourhood(t, width=40):
    c = np.float32(t)
    lo = c
    for _ in range(width):
        lo = _prev(lo)
    out = [lo]
    for _ in range(2 * width):
        out.append(np.nextafter(out[-1], np.float32(np.inf), dtype=np.float32))
    return out


def test_float32_cut_equals_the_float64_comparison():
    rng = np.random.default_rng(5)
    ts = list(np.abs(rng.standard_normal(200)) * 10.0 ** rng.integers(-6, 6, 200))
    ts += [float(np.float32(1.5)), float(np.float32(0.1)), 1e-30, 3.0e38, float(np.finfo(np.float32).tiny) / 4, 2.0 ** -149]
    for t in ts:
        for kind in ("hard", "soft", "semisoft"):
            c = cut(kind, t)
            for x in _neighbourhood(t):
                for sx in (x, -x):
                    if kind == "semisoft" and sx < 0:
                        # the reference tests x <= 2t, not |x| <= 2t (threshold_main.jl:60-70): every negative x passes it
                        assert zeroed_by_reference(kind, sx, t) == (abs(float(sx)) < t)
                    assert (np.abs(sx) <= c) == zeroed_by_reference(kind, sx, t), (kind, t, float(sx), float(c))


def test_stein_cut_is_sufficient():
    rng = np.random.default_rng(6)
    for t in list(np.abs(rng.standard_normal(300)) * 10.0 ** rng.integers(-5, 5, 300)) + [1e-20, 1e20]:
        c = cut("stein", t)
        assert float(c) <= 0.999 * t
        xs = [c, _prev(c), np.float32(0.5) * c, np.float32(1e-3) * c, np.float32(0.0)] + list(np.float32(rng.random(50)) * c)
        for x in xs:
            for sx in (x, -x):
                assert zeroed_by_reference("stein", sx, t), (t, float(sx))


def test_no_cut_for_nonpositive_thresholds():
    for kind in ("soft", "semisoft", "stein"):
        assert cut(kind, 0.0) < 0 and cut(kind, -1.0) < 0
    assert cut("hard", 0.0) == 0 and cut("hard", -1.0) < 0
    # t = +Inf (an estnoise that returns Inf): everything finite is zeroed; an infinite x must reach the exact path (-> NaN)
    for kind in ("soft", "semisoft", "stein"):
        assert cut(kind, float("inf")) == np.finfo(np.float32).max
    assert np.isinf(cut("hard", float("inf")))
