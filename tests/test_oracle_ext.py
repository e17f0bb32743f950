"""The oracle for SURVEY.md section 8(f) rows 3-4 (modwt, threshold, noise estimate, denoise).  The reference's
tests hold NO golden vectors for these ("parity unpinned", oracle/wl_oracle_ext.c header): the oracle is pinned on
the properties the reference's own tests assert (test/transforms.jl:325-344: imodwt(modwt(x)) ~ x, sizes, level
prefixes) plus independent numpy formulations of the same definitions."""
import numpy as np
import pytest

from conftest import rng_array


def doppler(n):
    """testfunction(n, "Doppler"), Util (util_main.jl): sqrt(t(1-t)) sin(2 pi 1.05 / (t + 0.05)), t = i/n"""
    t = np.arange(1, n + 1) / n
    return np.sqrt(t * (1 - t)) * np.sin(2 * np.pi * 1.05 / (t + 0.05))


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
def test_modwt_roundtrip_sizes_prefix(oracle, W, dtype, tol):
    wf = W.wavelet(W.WT.db4)
    rng = np.random.default_rng(1)
    for x in (rng.standard_normal(128), np.cumsum(rng.standard_normal(129))):     # test/transforms.jl:326-338
        x = x.astype(dtype)
        Wm = oracle.modwt(x, wf.qmf)
        assert Wm.shape == (len(x), oracle.maxmodwttransformlevels(len(x)) + 1) and Wm.dtype == dtype
        xb = oracle.imodwt(Wm, wf.qmf)
        assert np.abs(xb - x).max() <= tol * max(1, np.abs(x).max())
        Wl = oracle.modwt(x, wf.qmf, 4)                                            # :340-343
        assert np.array_equal(Wm[:, :3], Wl[:, :3])
        # unit energy (the reason for the 1/sqrt 2 tap scaling)
        e0, e1 = (x.astype(np.float64) ** 2).sum(), (Wm.astype(np.float64) ** 2).sum()
        assert abs(e0 - e1) <= 50 * tol * e0


def test_modwt_matches_numpy_definition(oracle, W):
    """W_j[t] = sum_n h[n] V_{j-1}[t - n 2^(j-1) mod N] with h = mirror(qmf)/sqrt 2, g = reverse(qmf)/sqrt 2"""
    for fname in ("haar", "db2", "db4", "sym5", "coif2"):
        q = np.asarray(W.wavelet(getattr(W.WT, fname)).qmf)
        g = q[::-1] / np.sqrt(2)
        h = q * (-1.0) ** np.arange(len(q)) / np.sqrt(2)
        x = rng_array((200,), np.float64, 5)
        V = x.copy()
        Wm = oracle.modwt(x, q, 5)
        for j in range(1, 6):
            Wj = sum(h[n] * np.roll(V, n * 2 ** (j - 1)) for n in range(len(q)))
            V = sum(g[n] * np.roll(V, n * 2 ** (j - 1)) for n in range(len(q)))
            assert np.allclose(Wm[:, j - 1], Wj, rtol=0, atol=1e-13)
        assert np.allclose(Wm[:, 5], V, rtol=0, atol=1e-13)


def test_modwt_argument_contract(oracle, W):
    q = W.wavelet(W.WT.db2).qmf
    x = rng_array((100,), np.float64, 0)
    assert oracle.maxmodwttransformlevels(100) == 6 and oracle.maxmodwttransformlevels(128) == 7
    with pytest.raises(oracle.OracleError) as e:
        oracle.modwt(x, q, 7)
    assert e.value.rc == -1
    with pytest.raises(oracle.OracleError) as e:
        oracle.modwt(x, q, 0)
    assert e.value.rc == -2


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_threshold_definitions(oracle, dtype):
    x = (rng_array((400,), np.float64, 3) * 2).astype(dtype)
    x[:5] = [0.0, 2.0, -2.0, 4.0, -4.0]
    for t in (2, 1.25):
        C = np.float64 if (isinstance(t, float) or dtype == np.float64) else np.float32
        xc, tc = x.astype(C), C(t)
        ax = np.abs(xc)
        exp = {
            "hard": np.where(ax <= tc, 0, xc),
            "soft": np.where(ax - tc < 0, 0, np.sign(xc) * (ax - tc)),
            "semisoft": np.where(xc <= 2 * tc, np.where(ax - tc < 0, 0, np.where(ax - tc - tc < 0, np.sign(xc) * (ax - tc) * 2, xc)), xc),
        }
        with np.errstate(divide="ignore", invalid="ignore"):
            sh = 1 - tc * tc / (xc * xc)
            exp["stein"] = np.where(sh < 0, 0, xc * sh)
        for kind, e in exp.items():
            got = oracle.threshold(x, kind, t)
            assert got.dtype == dtype
            assert np.array_equal(got, e.astype(dtype), equal_nan=True), (kind, t)
    assert np.array_equal(oracle.threshold(x, "pos"), np.where(x > 0, 0, x))
    assert np.array_equal(oracle.threshold(x, "neg"), np.where(x < 0, 0, x))
    for m in (0, 1, 17, 400, 1000):
        got = oracle.threshold(x, "biggest", m=m)
        mm = min(m, len(x))
        assert np.count_nonzero(got) <= mm
        keep = np.argsort(np.abs(x), kind="stable")[len(x) - mm:]
        e = np.zeros_like(x)
        e[keep] = x[keep]
        assert np.array_equal(got, e)
    with pytest.raises(oracle.OracleError):
        oracle.threshold(x, "hard", -1.0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_median_mad(oracle, dtype):
    for n in (1, 2, 3, 10, 11, 1000, 4097):
        v = rng_array((n,), dtype, n)
        assert oracle.median(v) == float(np.median(v.astype(dtype)))      # numpy: mean of the two middles; same value
        dev = np.abs(v - dtype(np.median(v)))
        assert oracle.mad(v) == float(np.median(dev))
    v = rng_array((10,), dtype, 0)
    v[3] = np.nan
    assert np.isnan(oracle.median(v))
    assert oracle.median(np.array([1, 1, 1, 1], dtype=dtype)) == 1.0


def test_circshift_add_rmul(oracle):
    for shape, sh in (((10,), [3]), ((10,), [-4]), ((6, 5), [2, -1]), ((4, 3, 5), [1, 2, -2]), ((7,), [0]), ((7,), [15])):
        a = rng_array(shape, np.float64, 1)
        assert np.array_equal(oracle.circshift(a, sh), np.roll(a, sh, axis=tuple(range(len(sh)))))
    y, z = rng_array((50,), np.float32, 1), rng_array((50,), np.float32, 2)
    assert np.array_equal(oracle.arrayadd(y, z), y + z)
    assert np.array_equal(oracle.rmul(y, 1 / 3), (y.astype(np.float64) * (1 / 3)).astype(np.float32))


def test_denoise_reduces_noise(oracle, W):
    """test/threshold.jl:12-21 only runs denoise; here: it must actually remove noise from the Doppler signal"""
    n = 256
    x0 = doppler(n)
    x = x0 + 0.05 * np.random.default_rng(0).standard_normal(n)
    wt = W.DEFAULT_WAVELET
    fwd = lambda a, L: oracle.dwt_filter(a, wt.qmf, L)
    inv = lambda a, L: oracle.dwt_filter(a, wt.qmf, L, fw=False)
    t_unit = W.VisuShrink(n).t
    sigma = oracle.noisest(x, fwd)
    assert 0.03 < sigma < 0.08
    y = oracle.denoise(x, fwd, inv, 6, "hard", t_unit)
    yti = oracle.denoise(x, fwd, inv, 6, "hard", t_unit, TI=True, nspin=8)
    e0, e1, e2 = np.linalg.norm(x - x0), np.linalg.norm(y - x0), np.linalg.norm(yti - x0)
    assert e1 < 0.8 * e0 and e2 < e1
    # wt === nothing: plain thresholding of the samples
    yn = oracle.denoise(x, None, None, 0, "hard", t_unit)
    assert np.array_equal(yn, oracle.threshold(x, "hard", oracle.noisest(x, None) * t_unit))
