"""The opt-in fused arithmetic mode (libwavelets_mi355x_fma.so, W.set_arithmetic("fused")) against SURVEY.md 8(c)'s stated
tolerances -- the default "exact" mode is covered bit for bit by the rest of the suite.

Bounds (reference = the Float64 oracle on the same inputs; L = levels):
    Float32:  ||y - ref||_2 / ||ref||_2 <= 1e-6 * sqrt(L)   and   max|y - ref| <= 1e-5 * max(1, ||ref||_inf)
    Float64:  ||y - ref||_2 / ||ref||_2 <= 1e-13 * sqrt(L)
    round trip idwt(dwt(x)) vs x: 1e-5 (Float32), 1e-12 (Float64)
    golden vectors (the reference's own files, /root/reference/test/transforms.jl:15-16): ||y - golden||_2 <= 1e-9 * sqrt(len)
The fused build compiles the same kernels with -ffp-contract=fast: same summation order, a*b+c may round once.
"""
import math

import numpy as np
import pytest

from conftest import golden, golden_cases, make_filter

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fused(gpu, W):
    W.set_arithmetic("fused")
    assert W.get_arithmetic() == "fused" and W._lib.load()._name.endswith("libwavelets_mi355x_fma.so")
    yield W
    W.set_arithmetic("exact")


def _check(y, ref, L, f64):
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    rel = np.linalg.norm((y - ref).ravel()) / np.linalg.norm(ref.ravel())
    if f64:
        assert rel <= 1e-13 * math.sqrt(max(L, 1)), rel
    else:
        assert rel <= 1e-6 * math.sqrt(max(L, 1)), rel
        assert np.abs(y - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), np.abs(y - ref).max()
    return rel


def test_mode_switch_is_explicit_and_changes_bits(gpu, W, oracle):
    """The two modes are two libraries; switching retires the contexts of the one being left; the fused result is NOT the exact
    one (otherwise the mode would be a no-op) but the same kernel family ran."""
    rng = np.random.default_rng(5)
    x = np.asfortranarray(rng.standard_normal((1024, 1024)).astype(np.float32))
    wt = W.wavelet(W.WT.db4)
    xd = W.to_device(x)
    ye = W.to_host(W.dwt(xd, wt, 5)); ke = W.last_kernel()
    assert np.array_equal(ye, oracle.dwt_filter(x, wt.qmf, 5))
    with W.arithmetic("fused"):
        assert W.get_arithmetic() == "fused"
        yf = W.to_host(W.dwt(xd, wt, 5)); kf = W.last_kernel()
    assert W.get_arithmetic() == "exact"
    assert ke == kf
    assert not np.array_equal(ye, yf)
    _check(yf, oracle.dwt_filter(x.astype(np.float64), wt.qmf, 5), 5, False)
    assert np.array_equal(W.to_host(W.dwt(xd, wt, 5)), ye)        # back on the exact library
    with pytest.raises(ValueError):
        W.set_arithmetic("fast-and-loose")


def test_fused_golden_vectors(fused, oracle):
    """the reference's 27-family golden files, 1-D (64) and 2-D (8 x 8), Float64, full depth, at the reference's own bound"""
    W = fused
    x1 = golden("filter1d_data.txt")
    x2 = np.asfortranarray(golden("filter2d_data.txt"))
    for fam, num, cls, vm in golden_cases():
        wt = make_filter(W, cls, vm)
        y1 = W.to_host(W.dwt(W.to_device(x1), wt))
        e1 = golden(f"filter1d_{fam}{num}.txt")
        assert np.linalg.norm(y1 - e1) <= 1e-9 * math.sqrt(x1.size), (fam, num)
        y2 = W.to_host(W.dwt(W.to_device(x2), wt))
        e2 = golden(f"filter2d_{fam}{num}.txt")
        assert np.linalg.norm((y2 - e2).ravel()) <= 1e-9 * math.sqrt(x2.size), (fam, num)


def test_fused_c1_c2_c4(fused, oracle):
    """C1 (db2 f64 2^20, L = 20), C2 (db4 f32 2^24, L = 24), C4 (cdf9/7 lifting f32 2^24, L = 24): forward against the
    Float64 oracle, inverse against the oracle's inverse, round trip"""
    W = fused
    rng = np.random.default_rng(11)
    u = rng.random(1 << 20)
    w2 = W.wavelet(W.WT.db2)
    yu = W.dwt(W.to_device(u), w2)
    ref = oracle.dwt_filter(u, w2.qmf)
    _check(W.to_host(yu), ref, 20, True)
    xr = W.to_host(W.idwt(yu, w2))
    assert np.linalg.norm(xr - u) / np.linalg.norm(u) <= 1e-12
    _check(W.to_host(W.idwt(W.to_device(ref), w2)), oracle.dwt_filter(ref, w2.qmf, fw=False), 20, True)

    v = rng.standard_normal(1 << 24).astype(np.float32)
    v64 = v.astype(np.float64)
    wt = W.wavelet(W.WT.db4)
    yv = W.dwt(W.to_device(v), wt)
    assert W.last_kernel() == "k_fwd1d_multi"
    ref = oracle.dwt_filter(v64, wt.qmf)
    _check(W.to_host(yv), ref, 24, False)
    assert np.linalg.norm(W.to_host(W.idwt(yv, wt)).astype(np.float64) - v64) / np.linalg.norm(v64) <= 1e-5
    _check(W.to_host(W.idwt(W.to_device(ref.astype(np.float32)), wt)), oracle.dwt_filter(ref.astype(np.float32).astype(np.float64), wt.qmf, fw=False), 24, False)

    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    yl = W.dwt(W.to_device(v), sch)
    refl = oracle.dwt_lifting(v64, sch)
    _check(W.to_host(yl), refl, 24, False)
    assert np.linalg.norm(W.to_host(W.idwt(yl, sch)).astype(np.float64) - v64) / np.linalg.norm(v64) <= 1e-5


def test_fused_c3_headline(fused, oracle):
    """C3: 2-D db4 8192 x 8192 f32 at L = 13 and L = 1, forward and inverse, element-wise bounds against the Float64 oracle;
    plus the 16-tap forward/inverse (the kernels the mode pays most on) and 2-D cdf9/7 lifting 4096 x 4096"""
    W = fused
    rng = np.random.default_rng(12)
    xh = np.asfortranarray(rng.standard_normal((8192, 8192)).astype(np.float32))
    x64 = xh.astype(np.float64)
    x = W.to_device(xh)
    nx = np.linalg.norm(x64.ravel())
    for name in ("db4", "sym8"):
        wt = W.wavelet(getattr(W.WT, name))
        for L in ((13, 1) if name == "db4" else (13,)):
            ref = oracle.dwt2d_filter_mt(x64, wt.qmf, L)
            y = W.dwt(x, wt, L)
            _check(W.to_host(y), ref, L, False)
            xr = W.to_host(W.idwt(y, wt, L)).astype(np.float64)
            assert np.linalg.norm((xr - x64).ravel()) / nx <= 1e-5, (name, L)
            del y, xr, ref
    del x, x64
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    xl = np.asfortranarray(rng.standard_normal((4096, 4096)).astype(np.float32))
    yl = W.dwt(W.to_device(xl), sch)
    _check(W.to_host(yl), oracle.dwt_lifting(xl.astype(np.float64), sch), 12, False)
    assert np.linalg.norm((W.to_host(W.idwt(yl, sch)) - xl).ravel().astype(np.float64)) / np.linalg.norm(xl.ravel().astype(np.float64)) <= 1e-5


def test_fused_c5_shard_and_other_entry_points(fused, oracle):
    """a C5 shard (2^16 x 2048 signals, db4, L = 16) on sampled columns; 3-D; wpt; modwt -- every family of the ABI answers in
    the fused library too"""
    W = fused
    import torch
    rng = np.random.default_rng(13)
    wt = W.wavelet(W.WT.db4)
    g = torch.Generator(device="cpu").manual_seed(3)
    xb = torch.randn(2048, 1 << 16, generator=g, dtype=torch.float32).to(gpu_dev()).t()
    yb = W.dwtc(xb, wt, 16)
    for j in (0, 1, 63, 64, 1023, 2047):
        col = xb[:, j].cpu().numpy().astype(np.float64)
        _check(yb[:, j].cpu().numpy(), oracle.dwt_filter(col, wt.qmf, 16), 16, False)
    del xb, yb
    c = np.asfortranarray(rng.standard_normal((128, 128, 128)).astype(np.float32))
    _check(W.to_host(W.dwt(W.to_device(c), wt, 5)), oracle.dwt_filter(c.astype(np.float64), wt.qmf, 5), 5, False)
    v = rng.standard_normal(1 << 18)
    tree = W.maketree(v.size, 6, "full")
    _check(W.to_host(W.wpt(W.to_device(v), wt, tree)), oracle.wpt_filter(v, wt.qmf, tree), 6, True)
    m = W.to_host(W.modwt(W.to_device(v), wt, 6))
    _check(m, oracle.modwt(v, wt.qmf, 6), 6, True)


def gpu_dev():
    import torch
    return torch.device("cuda", 0)
