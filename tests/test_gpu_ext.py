"""GPU parity for SURVEY.md section 8(f) rows 3-4: modwt/imodwt, threshold!, median/mad, circshift, and the whole
denoise pipeline, through the C ABI, BIT-EXACT against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from conftest import rng_array
from test_oracle_ext import doppler

pytestmark = pytest.mark.gpu


def host(W, t):
    import torch
    torch.cuda.synchronize()
    return W.to_host(t)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_modwt_bitexact(gpu, W, oracle, dtype):
    for n, Ls in ((128, (None, 1, 4)), (129, (None, 3)), (1000, (None, 9)), (1 << 16, (None, 5)), (1, ()), (2, (1,)), (3, (1,)), (8, (1, 3)), (12, (2,)),
                  (4, (1, 2))):
        x = np.cumsum(rng_array((n,), np.float64, n)).astype(dtype)
        for fname in ("db4", "haar", "db2", "sym5", "batt2", "coif6"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                we = oracle.modwt(x, wt.qmf, L)
                wg = W.modwt(W.to_device(x), wt, L)
                assert tuple(wg.shape) == we.shape and W.is_julia_layout(wg)
                assert np.array_equal(host(W, wg), we), (n, fname, L)
                xr = host(W, W.imodwt(W.to_device(we), wt))
                assert np.array_equal(xr, oracle.imodwt(we, wt.qmf)), (n, fname, L, "inv")
                with W.options(WL_MODWT_SMALL=0):        # levels 1-2 on the scalar kernels instead of the small-stride vector kernels
                    assert np.array_equal(host(W, W.modwt(W.to_device(x), wt, L)), we), (n, fname, L, "scalar")
                    assert np.array_equal(host(W, W.imodwt(W.to_device(we), wt)), xr), (n, fname, L, "scalar inv")
                if fname != "batt2":          # the Battle tables are not orthogonal: imodwt is only the adjoint there
                    assert np.abs(xr - x).max() <= (1e-4 if dtype == np.float32 else 1e-10) * max(1.0, np.abs(x).max())
    wt = W.wavelet(W.WT.db4)
    xd = W.to_device(rng_array((100,), dtype, 0))
    assert W.maxmodwttransformlevels(xd) == 6
    with pytest.raises(W.ArgumentError, match="Too many transform levels"):
        W.modwt(xd, wt, 7)
    with pytest.raises(W.ArgumentError, match="L must be >= 1"):
        W.modwt(xd, wt, 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_threshold_bitexact(gpu, W, oracle, dtype):
    kinds = {"hard": W.HardTH(), "soft": W.SoftTH(), "semisoft": W.SemiSoftTH(), "stein": W.SteinTH()}
    for n in (1, 5, 200, 4099, 1 << 18):
        x = (rng_array((n,), np.float64, n) * 2).astype(dtype)
        x[: min(5, n)] = np.array([0.0, 2.0, -2.0, 4.0, -4.0], dtype=dtype)[: min(5, n)]
        for t in (2, 0, 1.25, 0.0, np.float32(0.7)):
            for kind, TH in kinds.items():
                got = host(W, W.threshold(W.to_device(x), TH, t))
                assert np.array_equal(got, oracle.threshold(x, kind, t), equal_nan=True), (n, kind, t)
        assert np.array_equal(host(W, W.threshold(W.to_device(x), W.PosTH())), oracle.threshold(x, "pos"))
        assert np.array_equal(host(W, W.threshold(W.to_device(x), W.NegTH())), oracle.threshold(x, "neg"))
        for m in sorted({0, 1, n // 3, n - 1, n, n + 5}):
            if m < 0:
                continue
            got = host(W, W.threshold(W.to_device(x), W.BiggestTH(), m))
            assert np.array_equal(got, oracle.threshold(x, "biggest", m=m)), (n, m)
    # ties at the cut (equal magnitudes, both signs) and many zeros
    x = np.array([3, -1, 1, 0, 0, -1, 2, 1, -3, 0, 1, -1], dtype=dtype)
    for m in range(0, len(x) + 1):
        got = host(W, W.threshold(W.to_device(x), W.BiggestTH(), m))
        assert np.array_equal(got, oracle.threshold(x, "biggest", m=m)), m
    # in place on a Julia-layout matrix; the non-! form leaves its input alone
    a = rng_array((64, 32), dtype, 4)
    ad = W.to_device(a)
    out = W.threshold(ad, W.SoftTH(), 0.5)
    assert np.array_equal(host(W, ad), a) and np.array_equal(host(W, out), oracle.threshold(a, "soft", 0.5))
    assert W.threshold_(ad, W.HardTH(), 1) is ad and np.array_equal(host(W, ad), oracle.threshold(a, "hard", 1))
    with pytest.raises(AssertionError):
        W.threshold_(ad, W.HardTH(), -1.0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_median_mad_bitexact(gpu, W, oracle, dtype):
    for n in (1, 2, 3, 10, 11, 1000, 4095, 4096, 4097, 8191, 8192, 8193, (1 << 20) + 3, 1 << 21):
        v = rng_array((n,), dtype, n)
        assert W.median(W.to_device(v)) == oracle.median(v), n
        vd = W.to_device(v)
        assert W.mad_(vd) == oracle.mad(v), n
        assert np.array_equal(host(W, vd), np.abs(v - dtype(oracle.median(v))))       # mad! leaves |y - m| behind
    for v in (np.array([1, 1, 1, 1]), np.array([-2.0, -2.0, 5.0]), np.array([0.0, -0.0, 0.0, 1.0]), np.array([1e30, -1e30, 3, 4]),
              np.repeat(np.array([2.0, -7.0, 2.0]), 1000), np.array([-1.5, -0.25, -3.0, -0.5])):
        v = v.astype(dtype)
        assert W.median(W.to_device(v)) == oracle.median(v)
        assert W.mad_(W.to_device(v)) == oracle.mad(v)
    v = rng_array((100,), dtype, 1)
    v[17] = np.nan
    assert np.isnan(W.median(W.to_device(v)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_circshift_bitexact(gpu, W, oracle, dtype):
    for shape, sh in (((10,), [3]), ((10,), [-4]), ((6, 5), [2, -1]), ((4, 3, 5), [1, 2, -2]), ((7,), [0]), ((7,), [15]),
                      ((1024, 512), [7, 500]), ((64, 64, 64), [1, 0, 63])):
        a = rng_array(shape, dtype, 1)
        got = host(W, W.circshift(W.to_device(a), sh))
        assert np.array_equal(got, oracle.circshift(a, sh)) and np.array_equal(got, np.roll(a, sh, axis=tuple(range(len(sh)))))


def _oracle_denoise(oracle, W, x, wt, L, dnt, TI, nspin, sigma=None):
    kind = type(dnt.th).__name__[:-2].lower()
    if wt is None:
        return oracle.denoise(x, None, None, 0, kind, dnt.t)
    if isinstance(wt, W.GLS):
        fwd = lambda a, l: oracle.dwt_lifting(a, wt, l)
        inv = lambda a, l: oracle.dwt_lifting(a, wt, l, fw=False)
    else:
        fwd = lambda a, l: oracle.dwt_filter(a, wt.qmf, l)
        inv = lambda a, l: oracle.dwt_filter(a, wt.qmf, l, fw=False)
    return oracle.denoise(x, fwd, inv, L, kind, dnt.t, TI=TI, nspin=nspin, sigma=sigma)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_denoise_pipeline_bitexact(gpu, W, oracle, dtype):
    """the calls of test/threshold.jl:12-21, checked value for value"""
    n = 2 ** 8
    x0 = doppler(n)
    x = (x0 + 0.05 * np.random.default_rng(0).standard_normal(n)).astype(dtype)
    xd = W.to_device(x)
    vs = W.VisuShrink(n)
    wt = W.DEFAULT_WAVELET
    assert W.noisest(xd) == oracle.noisest(x, lambda a, l: oracle.dwt_filter(a, wt.qmf, l))
    # wt = nothing on lengths without a 2^L factor: the detail range follows the reference's round(Int, .) (ties to even):
    # n = 5, L = 1 -> Julia 4:5 (3.5 rounds to 4), n = 7 -> 4:7 (4.5 rounds to 4), n = 9, L = 2 -> 3:4
    for n5, L5, lohi in ((5, 1, (3, 5)), (7, 1, (3, 7)), (9, 2, (2, 4)), (11, 1, (5, 11)), (6, 2, (1, 3))):
        v = rng_array((n5,), dtype, n5)
        expect = oracle.mad(v[lohi[0]:lohi[1]].copy()) / 0.6745
        assert oracle.noisest(v, None, L5) == expect, (n5, L5)
        assert W.noisest(W.to_device(v), None, L5) == expect, (n5, L5)
    cases = [dict(TI=True), dict(TI=True, nspin=8), dict(TI=False), dict(wt=None), dict(TI=True, nspin=3),
             dict(wt=W.wavelet(W.WT.cdf97, W.WT.Lifting)), dict(wt=W.wavelet(W.WT.cdf97, W.WT.Lifting), TI=True, nspin=4),
             dict(dnt=W.VisuShrink(W.SoftTH(), 2.5), L=3), dict(wt=W.wavelet(W.WT.db4), dnt=W.VisuShrink(W.SteinTH(), 3.0))]
    for kw in cases:
        y = host(W, W.denoise(xd, **kw))
        w = kw.get("wt", wt)
        L = kw.get("L", min(W.maxtransformlevels(n), 6))
        e = _oracle_denoise(oracle, W, x, w, L, kw.get("dnt", vs), kw.get("TI", False), kw.get("nspin", (8,)))
        assert y.dtype == dtype and np.array_equal(y, e), kw
        if "dnt" not in kw and w is not None:          # (thresholding the raw samples is not a denoiser)
            assert np.linalg.norm(y - x0) < np.linalg.norm(x - x0)
    assert np.array_equal(host(W, xd), x)                        # denoise never modifies its input
    # 2-D, translation invariant, 8 x 8 spins (test/threshold.jl:20)
    a = rng_array((32, 32), dtype, 7)
    y = host(W, W.denoise(W.to_device(a), TI=True))
    e = _oracle_denoise(oracle, W, a, wt, min(W.maxtransformlevels(a), 6), W.VisuShrink(32), True, (8, 8))
    assert np.array_equal(y, e)
    y = host(W, W.denoise(W.to_device(a), TI=True, nspin=(2, 3)))
    assert np.array_equal(y, _oracle_denoise(oracle, W, a, wt, 5, W.VisuShrink(32), True, (2, 3)))
    with pytest.raises(W.ArgumentError, match="square/cube"):
        W.denoise(W.to_device(rng_array((32, 16), dtype, 1)))
    # the fused device-resident batch (wl_denoise_ti_filter): sizes that reach the plane-batched fast kernels, spins
    # processed in several groups (small buffer cap), a custom noise estimate handed over as a host scalar
    b = (doppler(512)[:, None] * doppler(512)[None, :] + 0.05 * np.random.default_rng(3).standard_normal((512, 512))).astype(dtype)
    bd = W.to_device(b)
    e = _oracle_denoise(oracle, W, b, wt, 6, W.VisuShrink(512), True, (4, 3))
    y = host(W, W.denoise(bd, TI=True, nspin=(4, 3)))
    assert W.last_kernel() == "denoise_ti_batch" and np.array_equal(y, e)
    W.set_option("WL_TI_WS_CAP_MB", 8)                           # 12 spins of 1 MiB: groups of a few spins
    assert np.array_equal(host(W, W.denoise(bd, TI=True, nspin=(4, 3))), e)
    W.clear_options()
    # ... and with the planes small enough to matter here sent through the batched fused-pair kernel (levels 1-2 and 3-4 in one
    # launch each: virtual shifts, hard threshold fused into the stores of both levels; by default only planes >= 2048^2 take it)
    W.set_option("WL_PAIR_BATCH_MIN", 0)
    assert np.array_equal(host(W, W.denoise(bd, TI=True, nspin=(4, 3))), e)
    es = _oracle_denoise(oracle, W, b, W.wavelet(W.WT.db4), 4, W.VisuShrink(W.SoftTH(), 2.0), True, (3, 5))
    ys = host(W, W.denoise(bd, W.wavelet(W.WT.db4), L=4, dnt=W.VisuShrink(W.SoftTH(), 2.0), TI=True, nspin=(3, 5)))
    assert np.array_equal(ys, es)
    eh = _oracle_denoise(oracle, W, b, W.wavelet(W.WT.db2), 2, W.VisuShrink(W.HardTH(), 1.5), True, (2, 2))
    yh = host(W, W.denoise(bd, W.wavelet(W.WT.db2), L=2, dnt=W.VisuShrink(W.HardTH(), 1.5), TI=True, nspin=(2, 2)))
    assert np.array_equal(yh, eh)                                # (L = 2: the pair is the last launch -- its approximation is thresholded too)
    W.clear_options()
    v = (doppler(4096) + 0.05 * np.random.default_rng(4).standard_normal(4096)).astype(dtype)
    e1 = _oracle_denoise(oracle, W, v, W.wavelet(W.WT.db4), 6, W.VisuShrink(W.SoftTH(), 2.0), True, (16,))
    y1 = host(W, W.denoise(W.to_device(v), W.wavelet(W.WT.db4), dnt=W.VisuShrink(W.SoftTH(), 2.0), TI=True, nspin=16))
    assert np.array_equal(y1, e1)
    # custom estnoise: the same estimate computed by the caller must give the same bits
    y2 = host(W, W.denoise(bd, TI=True, nspin=(4, 3), estnoise=lambda a, w: W.noisest(a, w)))
    assert np.array_equal(y2, e)
    # sizes that are not powers of two (the batch runs through the any-size per-axis kernels): 2-D image and a long line
    c = (0.5 * rng_array((200, 200), np.float64, 5)).astype(dtype)
    Lc = min(W.maxtransformlevels(c), 6)
    ec = _oracle_denoise(oracle, W, c, wt, Lc, W.VisuShrink(200), True, (3, 2))
    assert np.array_equal(host(W, W.denoise(W.to_device(c), TI=True, nspin=(3, 2))), ec)
    assert np.array_equal(host(W, W.denoise(W.to_device(c))), _oracle_denoise(oracle, W, c, wt, Lc, W.VisuShrink(200), False, (8, 8)))
    v3 = (0.5 * rng_array((3000,), np.float64, 6)).astype(dtype)
    L3 = min(W.maxtransformlevels(v3), 6)
    e3 = _oracle_denoise(oracle, W, v3, wt, L3, W.VisuShrink(3000), True, (5,))
    assert np.array_equal(host(W, W.denoise(W.to_device(v3), TI=True, nspin=5)), e3)
    # ---- edge cases of the translation-invariant branch (round-2 advice) ----
    # L = 0: dwt / idwt are copies, so every spin thresholds the shifted signal itself (odd length: no 2^L factor needed)
    for v0, nsp0 in ((rng_array((37,), dtype, 8), 5), (rng_array((24, 24), dtype, 9), (3, 2))):
        e0 = _oracle_denoise(oracle, W, v0, wt, 0, W.VisuShrink(W.HardTH(), 0.7), True, (nsp0,) if isinstance(nsp0, int) else nsp0, sigma=1.0)
        y0 = host(W, W.denoise(W.to_device(v0), L=0, dnt=W.VisuShrink(W.HardTH(), 0.7), TI=True, nspin=nsp0, estnoise=lambda a, w: 1.0))
        assert W.last_kernel() == "denoise_ti_batch" and np.array_equal(y0, e0) and not np.array_equal(y0, v0)
    # a vector with a longer nspin tuple: prod(nspin) = 16 spins shifted by 0 .. 15 (denoising.jl:38-42)
    y4 = host(W, W.denoise(xd, TI=True, nspin=(4, 4)))
    assert np.array_equal(y4, _oracle_denoise(oracle, W, x, wt, min(W.maxtransformlevels(n), 6), vs, True, (4, 4)))
    assert np.array_equal(y4, host(W, W.denoise(xd, TI=True, nspin=16)))
    # PosTH / NegTH take no threshold value: threshold!(xt, th, t) has no such method in the reference
    for th in (W.PosTH(), W.NegTH()):
        with pytest.raises(TypeError):
            W.denoise(xd, dnt=W.VisuShrink(th, 1.0), TI=True)
    # a custom estimator that returns NaN or a negative value trips @assert t >= 0 instead of being replaced silently
    for bad in (float("nan"), -1.0):
        with pytest.raises(AssertionError):
            W.denoise(xd, TI=True, estnoise=lambda a, w: bad)
    # ... while +Inf passes it, as in the reference: t = Inf thresholds every coefficient, the result is all zeros
    yinf = host(W, W.denoise(xd, TI=True, nspin=4, estnoise=lambda a, w: float("inf")))
    assert np.array_equal(yinf, np.zeros_like(yinf))
    # the plain denoise keeps the reference's own sequence (no shifted copy, no averaging): signed zeros survive
    xz = -np.abs(x) * 1e-30                     # every coefficient below the threshold; an all-negative-zero input for L = 0
    for Lz in (0, 3):
        ez = _oracle_denoise(oracle, W, xz, wt, Lz, vs, False, None)
        yz = host(W, W.denoise(W.to_device(xz), L=Lz))
        assert np.array_equal(yz, ez) and np.array_equal(np.signbit(yz), np.signbit(ez)), Lz


def test_ti_fused_thresholds_all_kinds(gpu, W, oracle):
    """The level kernels of the translation-invariant batch threshold the coefficients they store (round 4: every kind, not just
    hard): a Float32 cut per coefficient, threshold_one in Float64 for what survives the cut of soft / semisoft / Stein.  Both
    level kernels (k_fwd2d_lds per level; k_fwd2d_pair, two levels per launch), thresholds from "nothing survives" to "everything
    survives" (incl. t = 0: no cut at all), the pair as the last launch (its approximation is final), against the oracle and
    against the separate threshold pass."""
    rng = np.random.default_rng(11)
    b = (doppler(512)[:, None] * doppler(512)[None, :] + 0.05 * rng.standard_normal((512, 512))).astype(np.float32)
    bd = W.to_device(b)
    kinds = (W.SoftTH(), W.SemiSoftTH(), W.SteinTH(), W.HardTH())
    for pair in (0, 1):
        for th in kinds:
            for wname, L, nspin, t, sigma in (("db4", 4, (2, 2), 2.0, None), ("sym5", 3, (3, 2), 0.4, None), ("db2", 2, (2, 1), 1.0, None),
                                              ("db4", 2, (1, 2), 3.0, 0.05), ("sym5", 5, (2, 2), 0.0 if not isinstance(th, W.SteinTH) else 1e-3, 1.0)):
                wt = W.wavelet(getattr(W.WT, wname))
                dnt = W.VisuShrink(th, t)
                kw = dict(L=L, dnt=dnt, TI=True, nspin=nspin)
                if sigma is not None:
                    kw["estnoise"] = (lambda s: (lambda a, w: s))(sigma)
                e = _oracle_denoise(oracle, W, b, wt, L, dnt, True, nspin, sigma=sigma)
                W.clear_options()
                if pair:
                    W.set_option("WL_PAIR_BATCH_MIN", 0)
                y = host(W, W.denoise(bd, wt, **kw))
                assert W.last_kernel() == "denoise_ti_batch"
                assert np.array_equal(y, e, equal_nan=True), (pair, type(th).__name__, wname, L, t, int((y != e).sum()))
                W.set_option("WL_TI_FUSE_SOFT", 0)
                assert np.array_equal(host(W, W.denoise(bd, wt, **kw)), e, equal_nan=True), (pair, type(th).__name__, wname, "separate pass")
    W.clear_options()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_denoise_ti_lifting_batch_bitexact(gpu, W, oracle, dtype):
    """wl_denoise_ti_lifting (round 4): the translation-invariant branch of denoise for lifting schemes as one device-resident
    batch -- value for value the reference's per-spin sequence (denoising.jl:36-67) as restated by the oracle"""
    for sname in ("cdf97", "db2"):
        sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
        v = (doppler(2048) + 0.05 * np.random.default_rng(14).standard_normal(2048)).astype(dtype)
        for nsp, L, dnt in ((8, 6, W.VisuShrink(2048)), (5, 3, W.VisuShrink(W.SoftTH(), 1.5)), (3, 0, W.VisuShrink(W.HardTH(), 0.7))):
            e = _oracle_denoise(oracle, W, v, sch, L, dnt, True, (nsp,))
            y = host(W, W.denoise(W.to_device(v), sch, L=L, dnt=dnt, TI=True, nspin=nsp))
            assert W.last_kernel() == "denoise_ti_lifting" and np.array_equal(y, e), (sname, nsp, L)
        a = (doppler(256)[:, None] * doppler(256)[None, :] + 0.05 * np.random.default_rng(15).standard_normal((256, 256))).astype(dtype)
        for nsp, L, dnt in (((3, 2), 5, W.VisuShrink(256)), ((2, 2), 2, W.VisuShrink(W.SteinTH(), 1.2))):
            e = _oracle_denoise(oracle, W, a, sch, L, dnt, True, nsp)
            y = host(W, W.denoise(W.to_device(a), sch, L=L, dnt=dnt, TI=True, nspin=nsp))
            assert W.last_kernel() == "denoise_ti_lifting" and np.array_equal(y, e), (sname, nsp, L)
        # spins in several groups (small buffer cap) and a custom estimate handed over as a host scalar
        W.set_option("WL_TI_WS_CAP_MB", 1)
        e = _oracle_denoise(oracle, W, a, sch, 4, W.VisuShrink(256), True, (3, 3))
        assert np.array_equal(host(W, W.denoise(W.to_device(a), sch, L=4, TI=True, nspin=(3, 3))), e)
        W.clear_options()
        y2 = host(W, W.denoise(W.to_device(a), sch, L=4, TI=True, nspin=(3, 3), estnoise=lambda z, w: W.noisest(z, w)))
        assert np.array_equal(y2, e)
