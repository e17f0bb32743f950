"""GPU parity: the HIP path (called through the C ABI via the host mirror) against the CPU oracle
on the same seeded inputs.  The bar is BIT-EXACT equality for Float32 and Float64 (the kernels use
the reference's summation order with un-fused multiply/add); -0.0 == +0.0 is accepted."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden, golden_cases, make_filter, rng_array

pytestmark = pytest.mark.gpu


def dev(W, a):
    return W.to_device(a)


def host(W, t):
    import torch
    torch.cuda.synchronize()
    return W.to_host(t)


# ---- the reference's golden vectors, through the GPU --------------------------------------------
@pytest.mark.parametrize("fam,num,cls,vm", golden_cases())
def test_golden_vectors_on_gpu(gpu, W, oracle, fam, num, cls, vm):
    data = golden("filter1d_data.txt")
    data2 = golden("filter2d_data.txt")
    wt = make_filter(W, cls, vm)
    y = host(W, W.dwt(dev(W, data), wt))
    y2 = host(W, W.dwt(dev(W, data2), wt))
    assert np.linalg.norm(y - golden(f"filter1d_{fam}{num}.txt")) <= 1e-9 * 8
    assert np.linalg.norm(y2 - golden(f"filter2d_{fam}{num}.txt")) <= 1e-9 * 8
    assert np.array_equal(y, oracle.dwt_filter(data, wt.qmf))
    assert np.array_equal(y2, oracle.dwt_filter(data2, wt.qmf))
    # inverse (adjoint for the non-orthogonal Battle tables) is bit-identical too
    assert np.array_equal(host(W, W.idwt(dev(W, y), wt)), oracle.dwt_filter(y, wt.qmf, fw=False))
    assert np.array_equal(host(W, W.idwt(dev(W, y2), wt)), oracle.dwt_filter(y2, wt.qmf, fw=False))


def test_golden_nonsquare_on_gpu(gpu, W, oracle):
    data2 = golden("filter2d_nonsquare_data.txt")
    y2 = host(W, W.dwt(dev(W, data2), W.wavelet(W.WT.haar), 1))
    assert np.linalg.norm(y2 - golden("filter2d_nonsquare_Haar0.txt")) <= 1e-9 * np.sqrt(32)


# ---- filter bank, all ranks / sizes / levels -----------------------------------------------------
SHAPES = [(2,), (8,), (40,), (96,), (1024,), (4096,), (1 << 16,),
          (2, 2), (8, 8), (4, 8), (24, 40), (64, 64), (96, 32), (256, 128), (512, 512), (128, 1024), (1024, 512),
          (2, 2, 2), (8, 8, 8), (16, 8, 32), (32, 32, 32), (24, 8, 40)]
FILTERS = ["haar", "db2", "db4", "db3", "sym8", "coif6", "batt2"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_filter_fwd_inv_bitexact(gpu, W, oracle, dtype, shape):
    x = rng_array(shape, dtype, sum(shape))
    Lmax = W.maxtransformlevels(x)
    for fname in FILTERS:
        wt = W.wavelet(getattr(W.WT, fname))
        for L in sorted({0, 1, Lmax, max(Lmax - 1, 0)}):
            ye = oracle.dwt_filter(x, wt.qmf, L)
            y = host(W, W.dwt(dev(W, x), wt, L))
            assert y.dtype == dtype and y.shape == x.shape
            assert np.array_equal(y, ye), (fname, shape, L, np.abs(y - ye).max(), W.last_kernel())
            xr = host(W, W.idwt(dev(W, ye), wt, L))
            assert np.array_equal(xr, oracle.dwt_filter(ye, wt.qmf, L, fw=False)), (fname, shape, L, "inv")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fast_and_generic_paths_agree(gpu, W, oracle, dtype):
    """wl_ctx_set_path(1) forces the generic kernels; both families must give identical bits."""
    for shape, L in (((1 << 18,), 18), ((1 << 18,), 3), ((69632,), 12), ((69632,), 2), ((98304,), 15), ((1 << 20,), 5),
                     ((69632, 3), 0), ((1024, 1024), 10), ((512, 2048), 4), ((2048, 256), 8),
                     ((64, 64, 64), 6)):
        x = rng_array(shape, dtype, 99)
        for fname in ("db4", "db2", "haar", "db3", "sym4"):
            wt = W.wavelet(getattr(W.WT, fname))
            if L == 0:      # batched columns (dwtc), full depth of the column length
                fn = lambda t: W.dwtc(t, wt)
            else:
                fn = lambda t: W.dwt(t, wt, L)
            try:
                W.set_kernel_path(1)
                yg = host(W, fn(dev(W, x)))
                kg = W.last_kernel()
            finally:
                W.set_kernel_path(0)
            yf = host(W, fn(dev(W, x)))
            assert "generic" in kg
            assert np.array_equal(yg, yf), (shape, L, fname, W.last_kernel(), np.abs(yg - yf).max())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cube_axis_stream_kernels(gpu, W, oracle, dtype):
    """3-D levels assembled from k_fwd_axis_stream / k_inv_axis_stream / k_short_lines (wl_axis.hip): bit for bit
    against the oracle and the generic kernels, forward and inverse, every supported filter length."""
    for n, Ls in ((16, (1, 4)), (32, (1, 5)), (64, (1, 3, 6)), (128, (2,)), (256, (2,))):
        x = rng_array((n, n, n), dtype, n)
        for fname in ("db4", "haar", "db2", "db3", "db5", "sym8"):
            if n >= 128 and fname not in ("db4", "db5"):
                continue
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                xd = dev(W, x)
                y = host(W, W.dwt(xd, wt, L))
                kf = W.last_kernel()
                k1 = "k_fwd3d_one" if (n >= 128 and (len(wt.qmf) <= 8 or dtype == np.float32)) else "k_fwd_axis_stream"     # (round 6; 10 taps: Float32 only)
                if 4096 < n ** 3 <= 1 << 18:
                    k1 = "k_level3_lds"                                                            # (round 6: small levels in one launch)
                assert (kf == ("k_tail3" if n ** 3 <= 4096 else k1)) == (len(wt.qmf) <= 10), (n, fname, kf)
                if n <= 128:
                    ye = oracle.dwt_filter(x, wt.qmf, L)
                else:
                    try:
                        W.set_kernel_path(1)
                        ye = host(W, W.dwt(xd, wt, L))
                    finally:
                        W.set_kernel_path(0)
                assert np.array_equal(y, ye), (n, fname, L, np.abs(y - ye).max())
                xr = host(W, W.idwt(dev(W, ye), wt, L))
                ki = W.last_kernel()
                ki1 = "k_level3_lds" if 4096 < n ** 3 <= 1 << 18 else ("k_inv3d_one" if (n >= 256 and len(wt.qmf) <= 4) else "k_inv_axis_stream")
                assert (ki == ("k_tail3" if n ** 3 <= 4096 else ki1)) == (len(wt.qmf) <= 10), (n, fname, ki)
                if n <= 128:
                    xe = oracle.dwt_filter(ye, wt.qmf, L, fw=False)
                else:
                    try:
                        W.set_kernel_path(1)
                        xe = host(W, W.idwt(dev(W, ye), wt, L))
                    finally:
                        W.set_kernel_path(0)
                assert np.array_equal(xr, xe), (n, fname, L, "inv", np.abs(xr - xe).max())
                assert np.abs(xr - x).max() < (1e-4 if dtype == np.float32 else 1e-11)


@pytest.mark.parametrize("ppl", ["2", "4"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_inverse_2d_kernel(gpu, W, oracle, dtype, ppl):
    """k_inv2d_stream (dim-1 + dim-2 reconstruction of a 2-D level in one pass): bit for bit against the oracle;
    partial strips / chunks, non-square blocks, both lane widths, approximation taken from x (L = 1) and from the
    deeper reconstruction (L > 1)."""
    W.set_option("WL_INV2D_PPL", int(ppl))
    W.set_option("WL_TILE_INV", 0)            # (blocks <= 1024 would otherwise take the two-level LDS tile kernel)
    for shape, Ls in (((512, 512), (1, 2, 9)), ((1024, 2048), (1, 3)), ((2048, 512), (2, 9)), ((528, 96), (1, 4)),
                      ((4096, 16), (1,)), ((1000, 24), (1, 3)), ((128, 128), (1, 7)), ((256, 64), (1, 2)), ((136, 24), (1, 3)),
                      ((264, 528), (1, 3))):
        x = rng_array(shape, dtype, sum(shape))
        for fname in ("db4", "haar", "db2", "db3", "sym4", "db5"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                y = oracle.dwt_filter(x, wt.qmf, L)
                xr = host(W, W.idwt(dev(W, y), wt, L))
                if len(wt.qmf) <= 8 or shape[1] % 16 == 0:      # (blocks of <= 4096 elements: LDS tail kernel instead)
                    assert W.last_kernel() in (("k_inv2d_stream",) if shape[0] * shape[1] > 4096 else ("k_tail_inv", "k_tail2_inv")), (shape, L)
                assert np.array_equal(xr, oracle.dwt_filter(y, wt.qmf, L, fw=False)), (shape, fname, L)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_battle_filter_kernels(gpu, W, oracle, dtype):
    """Odd-length, 23..59-tap Battle-Lemarie filters (wt_main.jl:372-436): k_vl_lines_* (LDS-staged lines) and k_vl_axis_*
    (register window marching along the strided axis), 1-D / batched columns / 2-D incl. non-square blocks and axis lengths
    shorter than the filter, forward and inverse, bit for bit against the oracle; smaller levels fall to the generic kernels."""
    for fname, flen in (("batt2", 23), ("batt4", 41), ("batt6", 59)):
        wt = W.wavelet(getattr(W.WT, fname))
        assert len(wt.qmf) == flen
        for shape, Ls in (((4096,), (1, 3, 12)), ((1 << 16,), (16,)), ((1000 * 8,), (2,)), ((512, 512), (1, 2)), ((1024, 2048), (3,)),
                          ((2048, 64), (1,)), ((520, 96), (1,)), ((512, 32), (1,)), ((4096, 4096), (1,))):
            if dtype == np.float64 and shape in ((1024, 2048), (4096, 4096)):
                continue
            if shape == (4096, 4096) and flen != 59:
                continue
            x = rng_array(shape, dtype, flen + sum(shape))
            for L in Ls:
                ye = oracle.dwt2d_filter_mt(x, wt.qmf, L) if shape == (4096, 4096) else oracle.dwt_filter(x, wt.qmf, L)
                y = host(W, W.dwt(dev(W, x), wt, L))
                big = L == 1 and int(np.prod(shape)) > 16384      # smaller blocks are finished by the LDS tail kernels alone
                assert W.last_kernel() == "k_vl_lines" or not big, (fname, shape, L, W.last_kernel())
                assert np.array_equal(y, ye), (fname, shape, L, np.abs(y - ye).max())
                xr = host(W, W.idwt(dev(W, ye), wt, L))
                assert W.last_kernel() == "k_vl_lines" or not big, (fname, shape, L, W.last_kernel())
                xe = oracle.dwt2d_filter_mt(ye, wt.qmf, L, fw=False) if shape == (4096, 4096) else oracle.dwt_filter(ye, wt.qmf, L, fw=False)
                assert np.array_equal(xr, xe), (fname, shape, L, "inv")
        xm = rng_array((4096, 5), dtype, flen)
        assert np.array_equal(host(W, W.dwtc(dev(W, xm), wt, 4)), oracle.dwtc_filter(xm, wt.qmf, 4))
        ym = oracle.dwtc_filter(xm, wt.qmf, 4)
        assert np.array_equal(host(W, W.idwtc(dev(W, ym), wt, 4)), oracle.dwtc_filter(ym, wt.qmf, 4, fw=False))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_long_filter_kernels(gpu, W, oracle, dtype):
    """12..24-tap filters (db6..db10, sym6..sym10, coif4..coif8): k_long_lines (multi-lane halo) and the 32-slot
    ring of the axis kernels, 1-D / batched columns / 2-D, forward and inverse, bit for bit against the oracle."""
    filters = {"db6": 12, "db7": 14, "sym8": 16, "db9": 18, "db10": 20, "coif8": 24}
    for fname, flen in filters.items():
        wt = W.wavelet(getattr(W.WT, fname))
        assert len(wt.qmf) == flen
        for shape, Ls in (((4096,), (1, 3, 12)), ((1 << 16,), (16,)), ((1000 * 8,), (2,)), ((512, 512), (1, 2)), ((1024, 2048), (3,)),
                          ((2048, 64), (1,)), ((520, 96), (1,))):
            if dtype == np.float64 and shape == (1024, 2048):
                continue
            x = rng_array(shape, dtype, flen + sum(shape))
            for L in Ls:
                ye = oracle.dwt_filter(x, wt.qmf, L)
                y = host(W, W.dwt(dev(W, x), wt, L))
                big = int(np.prod(shape)) > 16384          # smaller blocks are finished by the LDS tail kernels alone
                kexp = "k_vl_lines" if flen == 24 else "k_long_lines"      # (24 taps: the register-window kernels at every size)
                # forward, Float32, 12..20 taps, rows a multiple of 256: one pass per level (wl_fwd2d_long.hip, its own test below)
                kfw = "k_fwd2d_lds_long" if (dtype == np.float32 and flen <= 20 and len(shape) == 2 and shape[0] % 256 == 0) else kexp
                # ... and its cache-resident levels (<= 1024 rows, two or more levels left) on the LDS tile kernel (round 5)
                if kfw == "k_fwd2d_lds_long" and max(shape) <= 1024 and min(shape) >= 128 and L >= 2:
                    kfw = "k_fwd2d_tile"
                assert W.last_kernel() == kfw or not big, (fname, shape, L, W.last_kernel())
                assert np.array_equal(y, ye), (fname, shape, L, np.abs(y - ye).max())
                xr = host(W, W.idwt(dev(W, ye), wt, L))
                # inverse, 12..20 taps, output rows a multiple of 256: one pass per level as well, both element types (wl_inv2d_long.hip, round 4)
                kinv = "k_inv2d_lds_long" if (flen <= 20 and len(shape) == 2 and shape[0] % 256 == 0 and shape[0] >= 256) else kexp
                # ... the last two levels of a block of <= 1024 rows in one launch of the two-level inverse tiles (round 5)
                if kinv == "k_inv2d_lds_long" and max(shape) <= 1024 and min(shape) >= 128 and L >= 2:
                    kinv = "k_inv2d_tile2"
                assert W.last_kernel() == kinv or not big, (fname, shape, L, W.last_kernel())
                assert np.array_equal(xr, oracle.dwt_filter(ye, wt.qmf, L, fw=False)), (fname, shape, L, "inv")
        xm = rng_array((4096, 5), dtype, flen)
        assert np.array_equal(host(W, W.dwtc(dev(W, xm), wt, 4)), oracle.dwtc_filter(xm, wt.qmf, 4))
        ym = oracle.dwtc_filter(xm, wt.qmf, 4)
        assert np.array_equal(host(W, W.idwtc(dev(W, ym), wt, 4)), oracle.dwtc_filter(ym, wt.qmf, 4, fw=False))


def test_inverse_fused_pair_kernel(gpu, W, oracle):
    """k_inv2d_pair (wl_inv.hip): levels 2 and 1 of a big Float32 reconstruction in one launch -- the level-2 output handed to the
    level-1 waves through an LDS column ring.  By default only from 4096 x 4096 upwards; WL_INV_PAIR_MIN = 0 brings it down to sizes
    the oracle checks bit for bit: every filter length it is instantiated for, square / wide / tall blocks, partial strips and
    chunks (1536 x 1056, 3072 x 160), odd and even depths (the pair always ends at level 1).  10 taps keep the single-level kernel."""
    W.set_option("WL_INV_PAIR_MIN", 0)
    W.set_option("WL_TILE_INV", 0)
    for shape, Ls in (((1024, 1024), (2, 3, 10)), ((2048, 512), (2, 4)), ((1024, 2048), (5,)), ((1536, 1056), (2,)), ((3072, 160), (2, 3))):
        x = rng_array(shape, np.float32, sum(shape) + 1)
        for fname in ("haar", "db2", "db3", "db4", "sym4", "sym5"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                xr = host(W, W.idwt(dev(W, x), wt, L))
                assert W.last_kernel() == ("k_inv2d_stream" if fname == "sym5" else "k_inv2d_pair"), (shape, fname, L, W.last_kernel())
                assert np.array_equal(xr, oracle.dwt_filter(x, wt.qmf, L, fw=False)), (shape, fname, L)
    # default dispatch at full size: the pair against two launches of the single-level kernel
    import torch
    xb = torch.randn(4096, 8192, generator=torch.Generator(device="cpu").manual_seed(3), dtype=torch.float32).cuda().t()
    W.clear_options()
    db4 = W.wavelet(W.WT.db4)
    a = W.idwt(xb, db4, 12)
    assert W.last_kernel() == "k_inv2d_pair"
    W.set_option("WL_INV_PAIR", 0)
    b = W.idwt(xb, db4, 12)
    assert W.last_kernel() == "k_inv2d_stream" and torch.equal(a, b)


def test_fused_level_pair_kernel(gpu, W, oracle):
    """k_fwd2d_stream2 (two 2-D levels per launch) is used from 4096^2 upwards by default; WL_FUSE2_MIN=0
    forces it on smaller blocks so that it can be checked bit for bit against the oracle and the generic
    kernels (odd/even L, non-square blocks, partial strips and chunks, every supported filter length)."""
    W.set_option("WL_FUSE2_MIN", 0)
    W.set_option("WL_LDS2D", 0)          # (the LDS-exchange kernel, tested below, takes these shapes by default)
    W.set_option("WL_M2D_MAX", 128)      # (... and the tile kernels everything up to 2048 x 2048)
    W.set_option("WL_TILE", 0)
    W.set_option("WL_TILEB", 0)
    for shape, Ls in (((512, 512), (2, 3, 9)), ((1024, 2048), (2, 5)), ((2048, 512), (4, 9)), ((528, 96), (2, 4)),
                      ((4096, 64), (2,))):
        x = rng_array(shape, np.float32, sum(shape))
        for fname in ("db4", "haar", "db2", "db3", "sym4"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                y = host(W, W.dwt(dev(W, x), wt, L))
                assert W.last_kernel() == "k_fwd2d_stream2", (shape, L, W.last_kernel())
                assert np.array_equal(y, oracle.dwt_filter(x, wt.qmf, L)), (shape, fname, L)
    # f64 has no fused kernel: same call goes through the single-level kernel
    x = rng_array((512, 512), np.float64, 3)
    y = host(W, W.dwt(dev(W, x), W.wavelet(W.WT.db4), 3))
    assert np.array_equal(y, oracle.dwt_filter(x, W.wavelet(W.WT.db4).qmf, 3))


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_lds_exchange_2d_kernel(gpu, W, oracle, mode):
    """k_fwd2d_lds (one fused 2-D level per launch, dim-1 pass through an LDS exchange): every workgroup shape
    (mode 0 = exact tiling with the halo helper wave where the row count allows it, 1..4 = overlapped strips of
    1..4 waves), every supported filter length, odd/even depths, non-square blocks, partial strips and chunks,
    row counts that are not multiples of a strip -- bit for bit against the oracle."""
    W.set_option("WL_LDS_MODE", mode)
    W.set_option("WL_FUSE2", 0)
    W.set_option("WL_M2D_MAX", 128)      # (the tile kernels would otherwise take everything up to 2048 x 2048)
    W.set_option("WL_TILE", 0)
    W.set_option("WL_TILEB", 0)
    shapes = (((512, 512), (1, 2, 3, 9)), ((1024, 2048), (2, 5)), ((2048, 512), (1, 4, 9)), ((528, 96), (1, 2, 4)), ((4096, 64), (1, 2)),
              ((256, 256), (1, 2)), ((1000, 64), (1, 2, 3)), ((272, 64), (1,)), ((768, 1024), (2, 3)), ((1280, 128), (1, 2)))
    for shape, Ls in shapes:
        x = rng_array(shape, np.float32, sum(shape) + mode)
        for fname in ("db4", "haar", "db2", "db3", "sym5"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                y = host(W, W.dwt(dev(W, x), wt, L))
                assert W.last_kernel() == "k_fwd2d_lds", (shape, L, W.last_kernel())
                ye = oracle.dwt_filter(x, wt.qmf, L)
                assert np.array_equal(y, ye), (shape, fname, L, mode, float(np.abs(y - ye).max()))


@pytest.mark.parametrize("tj", [32, 64, 128])
@pytest.mark.parametrize("wmain", [2, 4])
def test_fused_pair_2d_kernel(gpu, W, oracle, wmain, tj):
    """k_fwd2d_pair (wl_pair2d.hip: two fused 2-D levels per launch, the second one on a dedicated wave of every workgroup
    fed from an LDS column ring): strips of 512 / 1024 rows, every chunk length, every supported filter length (the
    level-2 lag and the number of steps past the chunk depend on it), odd / even depths (pairs followed by a single level),
    non-square blocks, one strip only (halo rows wrap onto the strip itself), chunks shorter than the others, blocks with
    64 columns -- bit for bit against the oracle.  Default dispatch uses it from 4096^2 upwards; WL_LDS_PAIR_MIN = 0 forces it."""
    W.set_option("WL_LDS_PAIR_MIN", 0)
    W.set_option("WL_PAIR_W", wmain)
    W.set_option("WL_TJ2", tj)
    W.set_option("WL_PAIR_WG_PER_CU", 0)  # (keep the requested chunk length)
    W.set_option("WL_M2D_MAX", 128)
    W.set_option("WL_TILE", 0)
    W.set_option("WL_TILEB", 0)
    shapes = (((512, 512), (2, 3, 9)), ((1024, 2048), (2, 5)), ((2048, 512), (2, 4, 9)), ((4096, 64), (2,)), ((1536, 160), (2, 3)),
              ((1024, 96), (2,)), ((512, 1056), (2, 4)))
    for shape, Ls in shapes:
        x = rng_array(shape, np.float32, sum(shape) + wmain + tj)
        for fname in ("db4", "haar", "db2", "db3", "sym5"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                y = host(W, W.dwt(dev(W, x), wt, L))
                assert W.last_kernel() == "k_fwd2d_pair", (shape, L, W.last_kernel())
                ye = oracle.dwt_filter(x, wt.qmf, L)
                if not np.array_equal(y, ye):
                    bad = np.argwhere(y != ye)
                    raise AssertionError((shape, fname, L, wmain, tj, len(bad), bad.min(axis=0).tolist(), bad.max(axis=0).tolist(),
                                          float(np.abs(y - ye).max())))
    # shapes the pair kernel declines (rows not a multiple of 512, columns not a multiple of 32) fall back to single levels
    for shape in ((768, 1024), (1024, 80)):
        x = rng_array(shape, np.float32, 5)
        wt = W.wavelet(W.WT.db4)
        y = host(W, W.dwt(dev(W, x), wt, 2))
        assert W.last_kernel() == "k_fwd2d_lds", (shape, W.last_kernel())
        assert np.array_equal(y, oracle.dwt_filter(x, wt.qmf, 2)), shape


@pytest.mark.parametrize("tj", [32, 64, 128])
def test_fused_pair_tile_launch(gpu, W, oracle, tj):
    """WL_FUSE4 = 1 (opt-in, round 6): levels l .. l+3 in ONE launch -- the fused pair plus, behind in-launch hand-over words, the
    64 x 64 two-level tiles of its approximation (fused_tile_role, wl_pair2d.hip: write-through stores, per-chunk progress words,
    sc1 loads, the last tile zeroes the words).  Every chunk length (a tile then depends on 3 .. 11 chunks), every filter length,
    depths that end inside / right after / well after the fused launch, non-square blocks; every call twice (the second call finds
    the hand-over words as the first one left them) -- bit for bit against the oracle."""
    W.set_option("WL_FUSE4", 1)
    W.set_option("WL_LDS_PAIR_MIN", 0)
    W.set_option("WL_TILEB_MIN", 0)
    W.set_option("WL_TJ2", tj)
    W.set_option("WL_PAIR_WG_PER_CU", 0)  # (keep the requested chunk length)
    W.set_option("WL_TILE", 0)            # (the cache-resident tiers would take these small blocks before the pair does)
    W.set_option("WL_M2D_MAX", 64)
    shapes = (((2048, 1024), (4, 5, 10)), ((1024, 2048), (4, 6)), ((512, 512), (4, 9)), ((1536, 768), (4, 5, 8)), ((4096, 512), (4, 7)))
    for shape, Ls in shapes:
        W.set_option("WL_TILEB_MAX", max(shape) // 4)
        x = rng_array(shape, np.float32, sum(shape) + tj)
        xd = dev(W, x)
        for fname in ("db4", "haar", "db2", "db3", "sym5"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                ye = oracle.dwt_filter(x, wt.qmf, L)
                for rep in range(2):
                    y = host(W, W.dwt(xd, wt, L))
                    assert W.last_kernel() == "k_fwd2d_pair_tile", (shape, L, W.last_kernel())
                    if not np.array_equal(y, ye):
                        bad = np.argwhere(y != ye)
                        raise AssertionError((shape, fname, L, tj, rep, len(bad), bad.min(axis=0).tolist(), bad.max(axis=0).tolist(),
                                              float(np.abs(y - ye).max())))
    # the headline shape: 30 back-to-back calls into the same output, compared with the two-launch chain of the default dispatch
    W.set_option("WL_LDS_PAIR_MIN", 1 << 24)
    W.set_option("WL_TILEB_MIN", 1 << 21)
    W.set_option("WL_TILEB_MAX", 2048)
    W.set_option("WL_TILE", 1)
    W.set_option("WL_M2D_MAX", 1024)
    W.set_option("WL_TJ2", 128)
    W.set_option("WL_PAIR_WG_PER_CU", 4)
    x = dev(W, rng_array((8192, 8192), np.float32, 77))
    wt = W.wavelet(W.WT.db4)
    W.set_option("WL_FUSE4", 0)
    yref = host(W, W.dwt(x, wt, 13))
    assert W.last_kernel() == "k_fwd2d_pair"
    W.set_option("WL_FUSE4", 1)
    for rep in range(30):
        y = W.dwt(x, wt, 13)
    assert W.last_kernel() == "k_fwd2d_pair_tile"
    assert np.array_equal(host(W, y), yref)
    # ... and the fused launch replays from a hipGraph (its hand-over words are reset by the launch itself, not by the host)
    import torch
    s = torch.cuda.Stream()
    xin = [rng_array((4096, 2048), np.float32, 90 + k) for k in range(3)]
    xg = dev(W, xin[0])
    yg = W.similar(xg)
    with torch.cuda.stream(s):
        for key, val in (("WL_FUSE4", 1), ("WL_LDS_PAIR_MIN", 0), ("WL_TILEB_MIN", 0), ("WL_TILEB_MAX", 1024), ("WL_TILE", 0), ("WL_M2D_MAX", 64)):
            W.set_option(key, val)                   # (options are per context = per stream)
        W.reserve_workspace(xg, 8, full=True)
        W.dwt_oop_(yg, xg, wt, 8)
    torch.cuda.synchronize()
    assert W.last_kernel() == "k_fwd2d_pair_tile"
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        W.dwt_oop_(yg, xg, wt, 8)
    for k in (1, 2, 0, 1):
        xg.copy_(dev(W, xin[k]))
        yg.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(W.to_host(yg), oracle.dwt_filter(xin[k], wt.qmf, 8)), k
    del graph


@pytest.mark.parametrize("tj", [16, 50, 128])
@pytest.mark.parametrize("wmain", [1, 2, 4])
def test_long_filter_single_pass_2d_kernel(gpu, W, oracle, wmain, tj):
    """k_fwd2d_lds_long (wl_fwd2d_long.hip): 12..20 taps in ONE pass per 2-D level (24-slot column ring, 16 / 20 / 24-row
    windows, guarded steps: chunk lengths that are not a multiple of the unrolled body), strips of 256 W rows, every filter
    length of the family, odd / even depths, non-square blocks, one strip only (halo rows wrap onto the strip itself), short
    last chunks -- bit for bit against the oracle; levels the kernel declines (rows not a multiple of 256) keep the two-pass kernels."""
    W.set_option("WL_LONG_W", wmain)
    W.set_option("WL_LONG_TJ", tj)
    W.set_option("WL_LONG_WG_PER_CU", 0)
    W.set_option("WL_LONG2D_MIN_ROWS", 256)
    W.set_option("WL_TILE_LONG", 0)                  # (round 5: these small blocks would otherwise take the LDS tile kernel, its own test below)
    shapes = (((512, 512), (1, 2, 3)), ((1024, 2048), (1, 2)), ((2048, 256), (1, 3)), ((256, 96), (1,)), ((768, 130), (1,)), ((1280, 1056), (1, 2)))
    for shape, Ls in shapes:
        x = rng_array(shape, np.float32, sum(shape) + wmain + tj)
        for fname in ("db6", "db7", "db8", "db9", "db10", "sym6", "sym8", "coif4", "coif6", "beyl"):
            wt = W.wavelet(getattr(W.WT, fname))
            assert len(wt.qmf) in (12, 14, 16, 18, 20), (fname, len(wt.qmf))
            for L in Ls:
                y = host(W, W.dwt(dev(W, x), wt, L))
                assert W.last_kernel() == "k_fwd2d_lds_long", (shape, fname, L, W.last_kernel())
                ye = oracle.dwt_filter(x, wt.qmf, L)
                if not np.array_equal(y, ye):
                    bad = np.argwhere(y != ye)
                    raise AssertionError((shape, fname, L, wmain, tj, len(bad), bad.min(axis=0).tolist(), bad.max(axis=0).tolist()))
    W.set_option("WL_LONG2D", 0)                     # the two-pass family on the same input gives the same bits
    x = rng_array((512, 512), np.float32, 4)
    wt = W.wavelet(W.WT.db8)
    y = host(W, W.dwt(dev(W, x), wt, 2))
    assert W.last_kernel() != "k_fwd2d_lds_long" and np.array_equal(y, oracle.dwt_filter(x, wt.qmf, 2))


@pytest.mark.parametrize("dtype,wmain,tp", [(np.float32, 1, 8), (np.float32, 1, 20), (np.float32, 1, 64), (np.float32, 2, 8), (np.float32, 2, 20),
                                            (np.float32, 2, 64), (np.float32, 4, 8), (np.float32, 4, 20), (np.float32, 4, 64),
                                            (np.float64, 1, 8), (np.float64, 1, 20), (np.float64, 2, 64), (np.float64, 4, 20)])
def test_long_filter_single_pass_2d_inverse_kernel(gpu, W, oracle, dtype, wmain, tp):
    """k_inv2d_lds_long (wl_inv2d_long.hip, round 4): ONE pass per 2-D inverse level -- 12..20 taps in Float32, 8..20 taps in
    Float64, and (enabled here) 8 / 10 taps in Float32 -- LDS exchange of the raw columns incl. the halo pairs that ride on
    wave 0, rings rounded up to a multiple of the request distance D, exit-guarded steps (chunk lengths that are not a multiple
    of the ring), strips of 256 W output rows, every filter length, odd / even depths, non-square blocks, a single strip (the
    halo wraps onto the strip itself), short last chunks, D = 1 ... 4 -- bit for bit against the oracle; the other tiers give the
    same bits."""
    W.set_option("WL_INVLONG_W", wmain)
    W.set_option("WL_INVLONG_TP", tp)
    W.set_option("WL_TILE_INV_LONG", 0)              # (round 5: the small blocks would otherwise take the two-level inverse tiles, own test below)
    W.set_option("WL_INVLONG_D", {8: 1, 20: 2, 64: 4}[tp])
    W.set_option("WL_INVLONG_PPL", {8: 2, 20: 1, 64: 0}[tp])       # pairs per lane: forced 2, forced 1 (Float32, W = 1), by size
    W.set_option("WL_INVLONG_WAVES_PER_CU", 0)
    W.set_option("WL_INVLONG2D_MIN_ROWS", 256)
    W.set_option("WL_INVLONG_FMIN", 8)
    W.set_option("WL_INVLONG_SHORT_MIN", 0)
    W.set_option("WL_INV_PAIR", 0)
    W.set_option("WL_TILE_INV", 0)
    shapes = (((512, 512), (1, 2, 3)), ((1024, 2048), (1, 2)), ((2048, 256), (1, 3)), ((256, 96), (1,)), ((768, 130), (1,)), ((1280, 1056), (1, 2)))
    for shape, Ls in shapes:
        x = rng_array(shape, dtype, sum(shape) + wmain + tp)
        for fname in ("db4", "sym5", "db6", "db7", "db8", "db9", "db10", "sym6", "sym8", "coif4", "coif6", "beyl"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                xr = host(W, W.idwt(dev(W, x), wt, L))
                assert W.last_kernel() == "k_inv2d_lds_long", (shape, fname, L, W.last_kernel())
                xe = oracle.dwt_filter(x, wt.qmf, L, fw=False)
                if not np.array_equal(xr, xe):
                    bad = np.argwhere(xr != xe)
                    raise AssertionError((shape, fname, L, wmain, tp, len(bad), bad.min(axis=0).tolist(), bad.max(axis=0).tolist()))
    W.set_option("WL_INVLONG2D", 0)
    x = rng_array((512, 512), dtype, 4)
    for fname in ("db8", "sym5"):
        wt = W.wavelet(getattr(W.WT, fname))
        xr = host(W, W.idwt(dev(W, x), wt, 2))
        assert W.last_kernel() != "k_inv2d_lds_long" and np.array_equal(xr, oracle.dwt_filter(x, wt.qmf, 2, fw=False))


@pytest.mark.parametrize("tj", [32, 128])
@pytest.mark.parametrize("wmain", [2, 4])
def test_float64_lds_exchange_and_pair_kernels(gpu, W, oracle, wmain, tj):
    """Float64 instances of the LDS-exchange level kernel (wl_fwd2d64.hip: two rows per lane, exact tiling) and of the fused
    pair (wl_pair2d64.hip): strips of 128 W rows, every supported filter length, odd / even depths, non-square blocks, a
    single strip (halo rows wrap onto the strip itself), short last chunks -- bit for bit against the oracle; shapes the
    kernels decline (rows not a multiple of 128) keep the round-1 kernels."""
    W.set_option("WL_PAIR_W64", wmain)
    W.set_option("WL_LDS_W", wmain)
    W.set_option("WL_TJ2", tj)
    W.set_option("WL_TJ", tj)
    W.set_option("WL_PAIR_WG_PER_CU", 0)
    W.set_option("WL_WAVES_PER_CU", 0)
    W.set_option("WL_WAVES_MIN", 0)
    W.set_option("WL_M2D_MAX", 128)
    W.set_option("WL_TILE", 0)
    W.set_option("WL_TILEB", 0)
    shapes = (((512, 512), (1, 2, 3, 9)), ((1024, 2048), (2, 5)), ((2048, 512), (1, 4)), ((256, 160), (1, 2)), ((768, 96), (2, 3)),
              ((1280, 1056), (1, 2)))
    for pairmin in (0, 1 << 62):
        W.set_option("WL_LDS_PAIR_MIN64", pairmin)
        for shape, Ls in shapes:
            x = rng_array(shape, np.float64, sum(shape) + wmain + tj)
            for fname in ("db4", "haar", "db2", "db3", "sym5"):
                wt = W.wavelet(getattr(W.WT, fname))
                for L in Ls:
                    y = host(W, W.dwt(dev(W, x), wt, L))
                    want = "k_fwd2d_pair64" if (pairmin == 0 and L >= 2 and shape[0] % 256 == 0) else "k_fwd2d_lds64"
                    assert W.last_kernel() == want, (shape, L, pairmin, W.last_kernel())
                    ye = oracle.dwt_filter(x, wt.qmf, L)
                    if not np.array_equal(y, ye):
                        bad = np.argwhere(y != ye)
                        raise AssertionError((shape, fname, L, wmain, tj, pairmin, len(bad), bad.min(axis=0).tolist(), bad.max(axis=0).tolist()))
    x = rng_array((320, 64), np.float64, 9)          # 320 rows: no strip of 128 divides it
    wt = W.wavelet(W.WT.db4)
    y = host(W, W.dwt(dev(W, x), wt, 2))
    assert W.last_kernel() not in ("k_fwd2d_lds64", "k_fwd2d_pair64") and np.array_equal(y, oracle.dwt_filter(x, wt.qmf, 2))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("nl3max", [0, 4096])
def test_tile_kernel(gpu, W, oracle, nl3max, dtype):
    """k_fwd2d_tile (1..3 fused levels of a cache-resident block per launch -- Float64: 1..2 --, one-sided halos, 64 x 64 tiles):
    every supported filter length, one / two / three levels per launch (WL_TILE_NL3_MAX), depths that end inside / after the tiled levels,
    non-square blocks, periodic wrap of the last tiles -- bit for bit against the oracle."""
    W.set_option("WL_TILE_NL3_MAX", nl3max)
    W.set_option("WL_TILE_MAX", 2048)
    W.set_option("WL_TILEB", 0)                     # (the variant without the staging buffer has its own test below)
    W.set_option("WL_LDS2D_MIN_ROWS", 1 << 20)      # keep the streaming kernel out: every level >= 128 goes to the tile kernel
    shapes = (((128, 128), (1, 2, 3, 7)), ((256, 256), (1, 2, 3, 4, 8)), ((512, 512), (2, 3, 5, 9)), ((1024, 1024), (3, 6)),
              ((256, 128), (1, 2, 3, 7)), ((128, 512), (2, 3)), ((2048, 2048), (2, 3)), ((192, 320), (1, 2, 3)))
    for shape, Ls in shapes:
        x = rng_array(shape, dtype, sum(shape) + nl3max)
        for fname in ("db4", "haar", "db2", "db3", "sym5"):
            if shape[0] >= 1024 and fname in ("db2", "db3"):
                continue
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                y = host(W, W.dwt(dev(W, x), wt, L))
                assert W.last_kernel() == "k_fwd2d_tile", (shape, L, W.last_kernel())
                ye = oracle.dwt_filter(x, wt.qmf, L)
                assert np.array_equal(y, ye), (shape, fname, L, nl3max, int((y != ye).sum()))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_inverse_tile_kernel(gpu, W, oracle, dtype):
    """k_inv2d_tile2 (two reconstruction levels of a block <= 1024 x 1024 per launch, 64 x 64 output tiles, recomputed one-sided
    halos in LDS): square and non-square blocks incl. the smallest (128: the tile's halo wraps around the 32-sample
    quarter extent), every filter length, depths that put the pair at the end, in the middle and after the tail kernel --
    bit for bit against the oracle and against the single-level streaming kernel."""
    for shape, Ls in (((128, 128), (2, 3, 7)), ((256, 256), (2, 8)), ((1024, 1024), (2, 4, 10)), ((256, 1024), (2, 3)), ((1024, 128), (2, 7)),
                      ((192, 320), (2, 3)), ((512, 512), (3, 9)), ((2048, 2048), (3, 11))):
        if dtype == np.float64 and shape == (2048, 2048):
            continue
        y = rng_array(shape, dtype, sum(shape))
        for fname in ("db4", "haar", "db2", "db3", "sym5"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                xe = oracle.dwt_filter(y, wt.qmf, L, fw=False)
                xr = host(W, W.idwt(dev(W, y), wt, L))
                # the LDS tail takes every level whose output is <= 4096 elements; the others are paired from the deep end
                l = sum(1 for q in range(1, L + 1) if (shape[0] >> (q - 1)) * (shape[1] >> (q - 1)) > 4096)
                last = None
                while l >= 1:
                    m, n = shape[0] >> max(l - 2, 0), shape[1] >> max(l - 2, 0)          # output of level l - 1
                    if l >= 2 and 128 <= m <= 1024 and 128 <= n <= 1024 and m % 64 == 0 and n % 64 == 0:
                        l, last = l - 2, "k_inv2d_tile2"
                    else:
                        l, last = l - 1, "k_inv2d_stream"
                if len(wt.qmf) <= 8:
                    assert W.last_kernel() == last, (shape, fname, L, W.last_kernel(), last)
                assert np.array_equal(xr, xe), (shape, fname, L, int((xr != xe).sum()))
                with W.options(WL_TILE_INV=0):
                    assert np.array_equal(host(W, W.idwt(dev(W, y), wt, L)), xe), (shape, fname, L, "streaming")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_any_even_size_tile_kernels(gpu, W, oracle, dtype):
    """k_fwd2d_gtile / k_inv2d_gtile (one 2-D level of ANY even extents per launch: 64 x 64 LDS tiles with a two-sided halo,
    periodic wrap by compare-and-subtract, clipped edges) -- the path of image-like shapes whose levels the streaming / tile /
    tail kernels decline: odd multiples of 2, 4, 6 ..., blocks smaller than a tile, blocks shorter than the filter, partial
    edge tiles; every filter length, forward and inverse, alone and below streaming levels -- bit for bit against the oracle
    and against the one-thread-per-output kernels they replace."""
    shapes = (((270, 480), (1,)), ((540, 960), (1, 2)), ((1080, 1920), (3,)), ((1000, 1000), (1, 3)), ((6, 10), (1,)), ((2, 2), (1,)),
              ((2, 130), (1,)), ((66, 62), (1,)), ((130, 258), (1,)), ((100, 36), (2,)), ((1260, 700), (2,)), ((24, 40), (3,)))
    for shape, Ls in shapes:
        x = rng_array(shape, dtype, sum(shape))
        for fname in ("db4", "haar", "db2", "db3", "sym5"):
            if shape[0] * shape[1] > 600000 and fname in ("db2", "db3"):
                continue
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                ye = oracle.dwt_filter(x, wt.qmf, L)
                y = host(W, W.dwt(dev(W, x), wt, L))
                kf = W.last_kernel()
                assert np.array_equal(y, ye), (shape, fname, L, kf, int((y != ye).sum()))
                xe = oracle.dwt_filter(ye, wt.qmf, L, fw=False)
                xr = host(W, W.idwt(dev(W, ye), wt, L))
                ki = W.last_kernel()
                assert np.array_equal(xr, xe), (shape, fname, L, ki, "inv")
                assert "generic" not in kf and "generic" not in ki, (shape, fname, L, kf, ki)
                with W.options(WL_GTILE=0):
                    assert np.array_equal(host(W, W.dwt(dev(W, x), wt, L)), ye), (shape, fname, L, "generic")
                    assert np.array_equal(host(W, W.idwt(dev(W, ye), wt, L)), xe), (shape, fname, L, "generic inv")
    for shape in ((270, 480), (130, 258), (1260, 700)):        # (blocks of <= 4096 elements go to the one-workgroup tails)
        x = rng_array(shape, dtype, 5)
        wt = W.wavelet(W.WT.db4)
        W.dwt(dev(W, x), wt, 1)
        assert W.last_kernel() == "k_fwd2d_gtile", (shape, W.last_kernel())
        W.idwt(dev(W, x), wt, 1)
        assert W.last_kernel() == "k_inv2d_gtile", (shape, W.last_kernel())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_any_axis_pass_kernels(gpu, W, oracle, dtype):
    """k_fwd_any / k_inv_any (one pass along any axis of a box of any even extent, four pairs per thread, compile-time taps):
    3-D volumes whose sides are not powers of two, batched columns of lengths that are not multiples of 8, lines with a large
    odd factor -- forward and inverse, bit for bit against the oracle and against the one-thread-per-output kernels."""
    cases = (((100, 100, 100), 2), ((60, 36, 20), 2), ((10, 6, 14), 1), ((2, 2, 2), 1), ((96, 96, 96), 5), ((240, 120, 40), 3), ((18, 50, 34), 1))
    for shape, L in cases:
        x = rng_array(shape, dtype, sum(shape))
        for fname in ("db4", "haar", "db3", "sym5", "db6", "sym8", "coif8"):
            if shape[0] >= 96 and fname in ("haar", "db3", "db6"):
                continue
            wt = W.wavelet(getattr(W.WT, fname))
            ye = oracle.dwt_filter(x, wt.qmf, L)
            y = host(W, W.dwt(dev(W, x), wt, L))
            kf = W.last_kernel()
            assert np.array_equal(y, ye), (shape, fname, L, kf, int((y != ye).sum()))
            xe = oracle.dwt_filter(ye, wt.qmf, L, fw=False)
            xr = host(W, W.idwt(dev(W, ye), wt, L))
            ki = W.last_kernel()
            assert np.array_equal(xr, xe), (shape, fname, L, ki, "inv")
            assert "generic" not in kf and "generic" not in ki, (shape, fname, kf, ki)
            with W.options(WL_ANYAXIS=0, WL_LEVEL3=0, WL_3D_ONE=0, WL_I3D_ONE=0):
                assert np.array_equal(host(W, W.dwt(dev(W, x), wt, L)), ye), (shape, fname, "generic")
                assert "generic" in W.last_kernel() or "tail" in W.last_kernel(), W.last_kernel()
                assert np.array_equal(host(W, W.idwt(dev(W, ye), wt, L)), xe), (shape, fname, "generic inv")
    wt = W.wavelet(W.WT.db4)
    for shape, L in (((44100, 6), 2), ((1004, 33), 2), ((50, 7), 1), ((100004,), 2), ((30,), 1)):
        x = rng_array(shape, dtype, 3)
        if len(shape) == 2:
            ye = oracle.dwtc_filter(x, wt.qmf, L)
            assert np.array_equal(host(W, W.dwtc(dev(W, x), wt, L)), ye), shape
            assert np.array_equal(host(W, W.idwtc(dev(W, ye), wt, L)), oracle.dwtc_filter(ye, wt.qmf, L, fw=False)), (shape, "inv")
        else:
            ye = oracle.dwt_filter(x, wt.qmf, L)
            assert np.array_equal(host(W, W.dwt(dev(W, x), wt, L)), ye), shape
            assert np.array_equal(host(W, W.idwt(dev(W, ye), wt, L)), oracle.dwt_filter(ye, wt.qmf, L, fw=False)), (shape, "inv")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_tail3_kernel(gpu, W, oracle, dtype):
    """k_tail3 (3-D: every remaining forward level / the deepest inverse levels of a power-of-two box <= 4096 elements in one
    workgroup, three LDS passes per level in the reference's order): cubes and non-cubic boxes, every depth, every
    supported filter length, alone and as the end of a larger transform -- bit for bit against the oracle and against
    the one-thread-per-output kernels it replaces."""
    for shape in ((16, 16, 16), (8, 8, 8), (4, 4, 4), (2, 2, 2), (16, 8, 4), (32, 8, 16), (4, 32, 32), (64, 8, 8)):
        x = rng_array(shape, dtype, sum(shape))
        Lmax = W.maxtransformlevels(x)
        for fname in ("db4", "haar", "db2", "db3", "sym5"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in range(1, Lmax + 1):
                ye = oracle.dwt_filter(x, wt.qmf, L)
                y = host(W, W.dwt(dev(W, x), wt, L))
                assert W.last_kernel() == "k_tail3", (shape, fname, L, W.last_kernel())
                assert np.array_equal(y, ye), (shape, fname, L)
                xe = oracle.dwt_filter(ye, wt.qmf, L, fw=False)
                xr = host(W, W.idwt(dev(W, ye), wt, L))
                assert W.last_kernel() == "k_tail3", (shape, fname, L, W.last_kernel())
                assert np.array_equal(xr, xe), (shape, fname, L, "inv")
                with W.options(WL_TAIL3=0):
                    assert np.array_equal(host(W, W.dwt(dev(W, x), wt, L)), ye), (shape, fname, L, "generic")
                    assert np.array_equal(host(W, W.idwt(dev(W, ye), wt, L)), xe), (shape, fname, L, "generic inv")
    for shape, L in (((64, 64, 64), 6), ((128, 32, 64), 5), ((32, 32, 32), 3)):
        x = rng_array(shape, dtype, 11)
        wt = W.wavelet(W.WT.db4)
        ye = oracle.dwt_filter(x, wt.qmf, L)
        assert np.array_equal(host(W, W.dwt(dev(W, x), wt, L)), ye), shape
        assert np.array_equal(host(W, W.idwt(dev(W, ye), wt, L)), oracle.dwt_filter(ye, wt.qmf, L, fw=False)), (shape, "inv")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_tail2_inverse_kernel(gpu, W, oracle, dtype):
    """k_tail2_inv (the deepest levels of a reconstruction whose output is a power-of-two block / line <= 16 KiB, one launch,
    mask wrap, in-place quadrant reuse): every supported filter length, every depth down to 2 x 2, square and non-square
    blocks, 1-D lines, batched lines, forced thread counts, as the start of a larger reconstruction -- bit for bit against
    the oracle and against the general inverse tail kernel."""
    cap = 4096
    for threads in (0, 64, 256):
        W.set_option("WL_TAIL2_THREADS", threads)
        for shape in ((64, 64), (32, 32), (64, 32), (16, 64), (8, 8), (4, 4), (2, 2), (128, 16), (2, 64), (64, 2)):
            if shape[0] * shape[1] > cap:
                continue
            y = rng_array(shape, dtype, sum(shape) + threads)
            Lmax = W.maxtransformlevels(y)
            for fname in ("db4", "haar", "db2", "db3", "sym5"):
                wt = W.wavelet(getattr(W.WT, fname))
                for L in sorted({1, 2, Lmax - 1, Lmax} - {0}):
                    if L > Lmax:
                        continue
                    xr = host(W, W.idwt(dev(W, y), wt, L))
                    assert W.last_kernel() == "k_tail2_inv", (shape, L, W.last_kernel())
                    assert np.array_equal(xr, oracle.dwt_filter(y, wt.qmf, L, fw=False)), (shape, fname, L, threads)
        for n in (2, 4, 8, 64, 512, 2048, 4096):
            if n > cap:
                continue
            y = rng_array((n,), dtype, n + threads)
            Lmax = W.maxtransformlevels(n)
            for fname in ("db4", "haar", "db3", "sym5"):
                wt = W.wavelet(getattr(W.WT, fname))
                for L in sorted({1, Lmax // 2, Lmax} - {0}):
                    xr = host(W, W.idwt(dev(W, y), wt, L))
                    assert W.last_kernel() == "k_tail2_inv", (n, L, W.last_kernel())
                    assert np.array_equal(xr, oracle.dwt_filter(y, wt.qmf, L, fw=False)), (n, fname, L, threads)
        ym = rng_array((256, 37), dtype, 5)                      # batched lines: one workgroup per line
        wt = W.wavelet(W.WT.db4)
        assert np.array_equal(host(W, W.idwtc(dev(W, ym), wt, 8)), oracle.dwtc_filter(ym, wt.qmf, 8, fw=False))
    W.set_option("WL_TAIL2_THREADS", 0)
    for shape, L in (((512, 512), 9), ((1024, 256), 8), ((1 << 16,), 16)):          # the tail feeds the streaming levels
        y = rng_array(shape, dtype, 3)
        for fname in ("db4", "sym5"):
            wt = W.wavelet(getattr(W.WT, fname))
            xe = oracle.dwt_filter(y, wt.qmf, L, fw=False)
            assert np.array_equal(host(W, W.idwt(dev(W, y), wt, L)), xe), (shape, fname)
            with W.options(WL_TAIL2=0):
                assert np.array_equal(host(W, W.idwt(dev(W, y), wt, L)), xe), (shape, fname, "general tail")


def test_tile_kernel_without_staging(gpu, W, oracle):
    """k_fwd2d_tileB (two fused levels, the first dim-2 pass straight from global memory: 33 KB of LDS, four workgroups per CU;
    default for blocks of 2^21 .. 2^22 elements, i.e. the 2048^2 level of C3): every filter length, even and odd depths (a pair
    followed by single levels / the tail), non-square blocks, the periodic wrap of the last tile row and column -- bit for bit."""
    W.set_option("WL_TILEB_MIN", 0)
    W.set_option("WL_LDS2D_MIN_ROWS", 1 << 20)
    W.set_option("WL_LDS_PAIR_MIN", 1 << 62)
    shapes = (((128, 128), (2, 3, 7)), ((256, 256), (2, 4, 8)), ((512, 512), (2, 3, 9)), ((1024, 1024), (2, 6)), ((256, 128), (2, 3)),
              ((128, 512), (2, 3)), ((2048, 2048), (2, 3, 11)), ((192, 320), (2, 3)), ((2048, 1024), (2,)))
    for shape, Ls in shapes:
        x = rng_array(shape, np.float32, sum(shape))
        for fname in ("db4", "haar", "db2", "db3", "sym5"):
            if shape[0] >= 1024 and fname in ("db2", "db3"):
                continue
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                y = host(W, W.dwt(dev(W, x), wt, L))
                assert W.last_kernel() == "k_fwd2d_tileB", (shape, L, W.last_kernel())
                ye = oracle.dwt_filter(x, wt.qmf, L)
                assert np.array_equal(y, ye), (shape, fname, L, int((y != ye).sum()))
    W.clear_options()
    # default dispatch: the 2048^2 block of an 8192^2 transform (after the fused pair) takes it
    x = rng_array((2048, 2048), np.float32, 3)
    wt = W.wavelet(W.WT.db4)
    y = host(W, W.dwt(dev(W, x), wt, 11))
    assert W.last_kernel() == "k_fwd2d_tileB" and np.array_equal(y, oracle.dwt_filter(x, wt.qmf, 11))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_tail2_kernel(gpu, W, oracle, dtype):
    """k_tail2_fwd (all remaining levels of a power-of-two block / line <= 16 KiB in one launch, mask wrap): every
    supported filter length, every depth down to 2 x 2 (lines shorter than the filter wrap several times), square and
    non-square blocks, 1-D lines, batched lines, forced thread counts -- bit for bit against the oracle and against the
    general tail kernel."""
    cap = 4096
    W.set_option("WL_NO_MULTI2D", 1)          # (64 x 64 blocks would otherwise take one pass of the tile kernel first)
    for threads in (0, 64, 256):
        W.set_option("WL_TAIL2_THREADS", threads)
        for shape in ((64, 64), (32, 32), (64, 32), (16, 64), (8, 8), (4, 4), (2, 2), (128, 16), (2, 64), (64, 2)):
            if shape[0] * shape[1] > cap:
                continue
            x = rng_array(shape, dtype, sum(shape) + threads)
            Lmax = W.maxtransformlevels(x)
            for fname in ("db4", "haar", "db2", "db3", "sym5"):
                wt = W.wavelet(getattr(W.WT, fname))
                for L in sorted({1, 2, Lmax - 1, Lmax} - {0}):
                    if L > Lmax:
                        continue
                    y = host(W, W.dwt(dev(W, x), wt, L))
                    assert W.last_kernel() == "k_tail2_fwd", (shape, L, W.last_kernel())
                    assert np.array_equal(y, oracle.dwt_filter(x, wt.qmf, L)), (shape, fname, L, threads)
        for n in (2, 4, 8, 64, 512, 2048, 4096):
            if n > cap:
                continue
            x = rng_array((n,), dtype, n + threads)
            Lmax = W.maxtransformlevels(n)
            for fname in ("db4", "haar", "db3", "sym5"):
                wt = W.wavelet(getattr(W.WT, fname))
                for L in sorted({1, Lmax // 2, Lmax} - {0}):
                    y = host(W, W.dwt(dev(W, x), wt, L))
                    assert W.last_kernel() == "k_tail2_fwd", (n, L, W.last_kernel())
                    assert np.array_equal(y, oracle.dwt_filter(x, wt.qmf, L)), (n, fname, L, threads)
        xm = rng_array((256, 37), dtype, 5)                      # batched lines: one workgroup per line
        wt = W.wavelet(W.WT.db4)
        assert np.array_equal(host(W, W.dwtc(dev(W, xm), wt, 8)), oracle.dwtc_filter(xm, wt.qmf, 8))
    W.set_option("WL_TAIL2", 0)                                  # the general tail kernel stays reachable and agrees
    x = rng_array((64, 64), dtype, 1)[:, : (64 if dtype == np.float32 else 32)]
    x = np.ascontiguousarray(x)
    wt = W.wavelet(W.WT.db4)
    y = host(W, W.dwt(dev(W, x), wt, 5))
    assert W.last_kernel() == "k_tail_fwd"
    assert np.array_equal(y, oracle.dwt_filter(x, wt.qmf, 5))


def test_randomized_shapes_near_dispatch_thresholds(gpu, W, oracle):
    """Seeded random sweep over shapes that sit on the eligibility boundaries of the fast kernels
    (strip/tile/chunk multiples, alignment, wrap) -- forward and inverse, against the oracle."""
    W.set_option("WL_FUSE2_MIN", int(0))
    rs = np.random.default_rng(2024)
    filters = ["haar", "db2", "db3", "db4", "db5", "sym4", "coif2"]
    n1d = [504, 512, 520, 1000, 1024, 2040, 4088, 16384, 16392, 20480, 32768, 49152, 65536 + 64]
    for n in n1d:
        x = rng_array((n,), np.float32 if rs.random() < 0.7 else np.float64, n)
        Lmax = W.maxtransformlevels(n)
        for _ in range(2):
            wt = W.wavelet(getattr(W.WT, filters[rs.integers(len(filters))]))
            L = int(rs.integers(1, Lmax + 1))
            ye = oracle.dwt_filter(x, wt.qmf, L)
            assert np.array_equal(host(W, W.dwt(dev(W, x), wt, L)), ye), (n, wt.name, L, W.last_kernel())
            assert np.array_equal(host(W, W.idwt(dev(W, ye), wt, L)), oracle.dwt_filter(ye, wt.qmf, L, fw=False)), (n, wt.name, L, "inv")
    dims = [240, 248, 256, 264, 496, 504, 512, 528, 544, 1024, 1040]
    cols = [16, 24, 32, 48, 64, 96, 128, 136, 256, 272]
    for _ in range(40):
        m, n = int(dims[rs.integers(len(dims))]), int(cols[rs.integers(len(cols))])
        if rs.random() < 0.3:
            m, n = n * 4, m // 4 * 2
        x = rng_array((m, n), np.float32 if rs.random() < 0.75 else np.float64, m * 7 + n)
        Lmax = W.maxtransformlevels(x)
        if Lmax == 0:
            continue
        wt = W.wavelet(getattr(W.WT, filters[rs.integers(len(filters))]))
        L = int(rs.integers(1, Lmax + 1))
        ye = oracle.dwt_filter(x, wt.qmf, L)
        assert np.array_equal(host(W, W.dwt(dev(W, x), wt, L)), ye), ((m, n), x.dtype, wt.name, L, W.last_kernel())
        assert np.array_equal(host(W, W.idwt(dev(W, ye), wt, L)), oracle.dwt_filter(ye, wt.qmf, L, fw=False)), ((m, n), wt.name, L, "inv")
    # lifting lines around the stream / tail thresholds
    for n in (504, 512, 520, 8192, 8200, 16384, 16392, 32768 + 8, 65536):
        x = rng_array((n,), np.float32 if n % 16 else np.float64, n)
        sch = W.wavelet(getattr(W.WT, ("cdf97", "db2", "haar")[n % 3]), W.WT.Lifting)
        L = int(rs.integers(1, W.maxtransformlevels(n) + 1))
        ye = oracle.dwt_lifting(x, sch, L)
        assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (n, sch.name, L, W.last_kernel())
        t = dev(W, ye)
        W.idwt_(t, sch, L)
        assert np.array_equal(host(W, t), oracle.dwt_lifting(ye, sch, L, fw=False)), (n, sch.name, L, "inv in place")
    # batched columns with leading dimension / line counts around the slab and alignment rules
    for (n, ns) in ((520, 3), (4096, 5), (16392, 2), (32768, 9)):
        x = rng_array((n, ns), np.float32, n + ns)
        wt = W.wavelet(W.WT.db4)
        L = W.maxtransformlevels(n)
        ye = oracle.dwtc_filter(x, wt.qmf, L)
        assert np.array_equal(host(W, W.dwtc(dev(W, x), wt, L)), ye), (n, ns)
        assert np.array_equal(host(W, W.idwtc(dev(W, ye), wt, L)), oracle.dwtc_filter(ye, wt.qmf, L, fw=False))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_multi_level_line_kernel_every_depth_per_launch(gpu, W, oracle, dtype):
    """k_fwd1d_multi: 1 ... 8 levels per launch (round 4 raised the cap for lines of a few MiB from 6 to 8: the halo
    (F-2)(2^NL - 1) then exceeds the tile), every filter length of the fast family, tile lengths from 1 KiB up, lines from 2^13
    to 2^20 -- bit for bit against the oracle."""
    for n, Ls in ((1 << 13, (1, 5, 13)), (1 << 16, (8, 16)), (1 << 18, (3, 18)), (1 << 20, (8, 20)), (3 << 14, (6, 14))):
        x = rng_array((n,), dtype, n % 1000 + 3)
        for fname in ("haar", "db2", "db3", "db4", "sym5"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in Ls:
                ye = oracle.dwt_filter(x, wt.qmf, L)
                for per, ts in ((8, 0), (7, 0), (5, 2048), (3, 1024), (1, 0), (8, 256)):
                    with W.options(WL_NL_SMALL_MAX=per, WL_NLMAX=per, WL_TS=ts):
                        y = host(W, W.dwt(dev(W, x), wt, L))
                        assert W.last_kernel() in ("k_fwd1d_multi", "k_tail2_fwd", "k_tail_fwd"), W.last_kernel()
                        assert np.array_equal(y, ye), (n, fname, L, per, ts, int((y != ye).sum()))


# ---- lifting ----------------------------------------------------------------------------------------
LSHAPES = [(2,), (4,), (8,), (40,), (1024,), (1 << 15,), (2, 2), (8, 8), (32, 32), (96, 96), (256, 256), (512, 512), (1024, 1024), (576, 576),
           (4, 4, 4), (8, 8, 8), (16, 16, 16), (24, 24, 24), (32, 32, 32), (64, 64, 64)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", LSHAPES, ids=lambda s: "x".join(map(str, s)))
def test_lifting_fwd_inv_bitexact(gpu, W, oracle, dtype, shape):
    x = rng_array(shape, dtype, 5 + sum(shape))
    Lmax = W.maxtransformlevels(x)
    for sname in ("cdf97", "db2", "haar", "db1"):
        sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
        for L in sorted({0, 1, Lmax}):
            ye = oracle.dwt_lifting(x, sch, L)
            y = host(W, W.dwt(dev(W, x), sch, L))
            assert np.array_equal(y, ye), (sname, shape, L, np.abs(y - ye).max())
            # in-place entry point dwt!(y, scheme, L)
            t = dev(W, x)
            r = W.dwt_(t, sch, L)
            assert r is t and np.array_equal(host(W, t), ye)
            xr = host(W, W.idwt(dev(W, ye), sch, L))
            assert np.array_equal(xr, oracle.dwt_lifting(ye, sch, L, fw=False)), (sname, shape, L, "inv")
            t = dev(W, ye)
            W.idwt_(t, sch, L)
            assert np.array_equal(host(W, t), xr)


def test_lifting_cubes_fast_vs_generic(gpu, W, oracle):
    """3-D lifting through k_lift_axis_stream + k_lift_short_lines: against the oracle (128^3) and against the generic
    kernels (256^3), forward and inverse."""
    for n, L in ((128, 7), (128, 2), (256, 3)):
        x = rng_array((n, n, n), np.float32, n)
        for sname in ("cdf97", "db2", "haar"):
            sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
            y = host(W, W.dwt(dev(W, x), sch, L))
            assert "k_lift_short_lines" in W.last_kernel()
            if n <= 128:
                ye = oracle.dwt_lifting(x, sch, L)
                xe = oracle.dwt_lifting(ye, sch, L, fw=False)
            else:
                try:
                    W.set_kernel_path(1)
                    ye = host(W, W.dwt(dev(W, x), sch, L))
                    assert "generic" in W.last_kernel()
                    xe = host(W, W.idwt(dev(W, ye), sch, L))
                finally:
                    W.set_kernel_path(0)
            assert np.array_equal(y, ye), (n, sname, L)
            assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (n, sname, L, "inv")


def test_lifting_2d_tile_kernel(gpu, W, oracle):
    """k_lift2d_tile_fwd / k_lift2d_tile_inv (wl_lift_tile.hip): one 2-D lifting level of a block of 128 ... 2048 rows (a multiple of
    64) as 64 x 64 tiles with the scheme's dependency cone recomputed per segment -- every scheme shape, both element types, sizes
    with 2 ... 64 tiles per side (and 4096 with the size limit raised), forward and inverse, bit for bit against the oracle; the
    level-1 call in place must not take it (it reads what it overwrites)."""
    for n, Ls, tmax in ((128, (1, 2), 0), (192, (1,), 0), (256, (2, 8), 0), (320, (2,), 0), (1024, (3,), 0), (2048, (1, 11), 0), (4096, (2,), 4096)):
        for dtype in (np.float32, np.float64):
            if n >= 2048 and dtype == np.float64:
                continue
            x = rng_array((n, n), dtype, n + 5)
            for sname in ("cdf97", "db2", "haar"):
                if n == 4096 and sname != "cdf97":
                    continue
                sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
                for L in Ls:
                    if tmax:
                        W.set_option("WL_LIFT_TILE_MAX", tmax)
                    ye = oracle.dwt_lifting(x, sch, L)
                    y = host(W, W.dwt(dev(W, x), sch, L))
                    if n <= 2048 and n != 320:        # (320 -> 160: the second level is not a multiple of 64 and marches)
                        assert W.last_kernel() == "k_lift2d_tile", W.last_kernel()
                    assert np.array_equal(y, ye), (n, sname, L, dtype, np.abs(y - ye).max())
                    xe = oracle.dwt_lifting(ye, sch, L, fw=False)
                    xr = host(W, W.idwt(dev(W, ye), sch, L))
                    if n == 128 and dtype == np.float32:   # (round 4: the whole 128 x 128 inverse is one LDS-tail launch)
                        assert W.last_kernel() == "k_tail_lift2d_lds", W.last_kernel()
                    elif n <= 2048 and n != 320:
                        assert W.last_kernel() == "k_lift2d_tile", W.last_kernel()
                    assert np.array_equal(xr, xe), (n, sname, L, dtype, "inv")
                    if n == 256:
                        t = dev(W, x)
                        W.dwt_(t, sch, L)
                        assert np.array_equal(host(W, t), ye), (n, sname, L, dtype, "fwd in place")
                        t = dev(W, ye)
                        W.idwt_(t, sch, L)
                        assert np.array_equal(host(W, t), xe), (n, sname, L, dtype, "inv in place")
                    if tmax:
                        W.set_option("WL_LIFT_TILE_MAX", 2048)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("tp", ["64", "24", ""])
def test_lifting_2d_axis_stream_kernel(gpu, W, oracle, tp, fused):
    """2-D lifting levels: k_lift2d_fwd / k_lift2d_inv (both passes of a level in one kernel: register cascade along
    dim 2 + DPP lifting across the lanes along dim 1) and, with the fused kernels switched off, k_lift_axis_stream +
    the line kernels: every chunk length, every scheme shape, forward and inverse, bit for bit against the oracle."""
    if tp:
        W.set_option("WL_LIFT_TP", int(tp))
    if not fused:
        W.set_option("WL_NO_LIFT2D_FUSED", int(1))
    W.set_option("WL_LIFT_TILE", 0)          # (blocks of <= 2048 rows take the tile kernel by default: test_lifting_2d_tile_kernel)
    expect = "k_lift2d" if fused else "k_lift_axis_stream"
    for n, Ls in ((512, (1, 3)), (1024, (2,)), (576, (1,)), (2048, (1, 11))):
        for dtype in (np.float32, np.float64):
            if n == 2048 and dtype == np.float64:
                continue
            x = rng_array((n, n), dtype, n)
            for sname in ("cdf97", "db2", "haar"):
                sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
                for L in Ls:
                    ye = oracle.dwt_lifting(x, sch, L)
                    y = host(W, W.dwt(dev(W, x), sch, L))
                    assert expect in W.last_kernel(), W.last_kernel()
                    assert np.array_equal(y, ye), (n, sname, L, dtype, np.abs(y - ye).max())
                    xe = oracle.dwt_lifting(ye, sch, L, fw=False)
                    xr = host(W, W.idwt(dev(W, ye), sch, L))
                    assert expect in W.last_kernel(), W.last_kernel()
                    assert np.array_equal(xr, xe), (n, sname, L, dtype, "inv")
                    if n == 512:            # in place: level 1 must not read what it is overwriting
                        t = dev(W, x)
                        W.dwt_(t, sch, L)
                        assert np.array_equal(host(W, t), ye), (n, sname, L, dtype, "fwd in place")
                        t = dev(W, ye)
                        W.idwt_(t, sch, L)
                        assert np.array_equal(host(W, t), xe), (n, sname, L, dtype, "inv in place")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lifting_lines_fast_paths(gpu, W, oracle, dtype):
    """Fused lifting kernels (stream levels + LDS tail), forward / inverse / in place / batched, and
    the generic family forced through wl_ctx_set_path(1): all bit-identical to the oracle."""
    for n, Ls in (((1 << 16), (16, 1, 3)), (3 << 15, (15, 2)), ((1 << 18), (18, 5))):
        x = rng_array((n,), dtype, n % 1000)
        for sname in ("cdf97", "db2", "haar"):
            sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
            for L in Ls:
                ye = oracle.dwt_lifting(x, sch, L)
                y = host(W, W.dwt(dev(W, x), sch, L))
                assert "generic" not in W.last_kernel(), W.last_kernel()
                assert np.array_equal(y, ye), (n, sname, L, "fwd", W.last_kernel())
                t = dev(W, x)
                W.dwt_(t, sch, L)                                   # in place
                assert np.array_equal(host(W, t), ye), (n, sname, L, "fwd in place")
                xe = oracle.dwt_lifting(ye, sch, L, fw=False)
                assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (n, sname, L, "inv")
                t = dev(W, ye)
                W.idwt_(t, sch, L)
                assert np.array_equal(host(W, t), xe), (n, sname, L, "inv in place")
                try:
                    W.set_kernel_path(1)
                    yg = host(W, W.dwt(dev(W, x), sch, L))
                    assert "generic" in W.last_kernel()
                finally:
                    W.set_kernel_path(0)
                assert np.array_equal(yg, ye)
    # batched columns: 24 signals of 2^15 samples
    xb = rng_array((1 << 15, 24), dtype, 77)
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    for L in (15, 2):
        ye = oracle.dwtc_lifting(xb, sch, L)
        assert np.array_equal(host(W, W.dwtc(dev(W, xb), sch, L)), ye)
        assert np.array_equal(host(W, W.idwtc(dev(W, ye), sch, L)), oracle.dwtc_lifting(ye, sch, L, fw=False))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lifting_register_tail(gpu, W, oracle, dtype):
    """k_tail_lift_reg (every remaining forward lifting level of a power-of-two line in one wave's registers, rotating
    DPP neighbours, ds_bpermute below 64 pairs): every line length from 2 up to the hand-over size, every depth class,
    the three known scheme shapes, in place, batched, as the end of a long transform -- bit for bit against the oracle and
    against the LDS tail it replaces."""
    nmax = 4096 if dtype == np.float32 else 2048
    n = 2
    while n <= nmax:
        x = rng_array((n,), dtype, n)
        Lmax = W.maxtransformlevels(n)
        for sname in ("cdf97", "db2", "haar"):
            sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
            for L in sorted({1, 2, Lmax // 2, Lmax - 1, Lmax} - {0}):
                if L > Lmax:
                    continue
                ye = oracle.dwt_lifting(x, sch, L)
                y = host(W, W.dwt(dev(W, x), sch, L))
                assert W.last_kernel() == "k_tail_lift_reg", (n, sname, L, W.last_kernel())
                assert np.array_equal(y, ye), (n, sname, L)
                t = dev(W, x)
                W.dwt_(t, sch, L)
                assert np.array_equal(host(W, t), ye), (n, sname, L, "in place")
                # the inverse register tail (k_tail_lift_reg_inv) takes the levels whose deepest approximation is <= 64 samples
                xe = oracle.dwt_lifting(ye, sch, L, fw=False)
                xr = host(W, W.idwt(dev(W, ye), sch, L))
                if (n >> L) <= 64:
                    assert W.last_kernel() == "k_tail_lift_reg_inv", (n, sname, L, W.last_kernel())
                assert np.array_equal(xr, xe), (n, sname, L, "inv")
                t = dev(W, ye)
                W.idwt_(t, sch, L)
                assert np.array_equal(host(W, t), xe), (n, sname, L, "inv in place")
        n *= 2
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    for shape, L in (((1024, 40), 10), ((4096 if dtype == np.float32 else 2048, 5), 7), ((64, 33), 6), ((1 << 15, 24), 15)):
        xb = rng_array(shape, dtype, shape[1])
        assert np.array_equal(host(W, W.dwtc(dev(W, xb), sch, L)), oracle.dwtc_lifting(xb, sch, L)), shape
        yb = oracle.dwtc_lifting(xb, sch, L)
        assert np.array_equal(host(W, W.idwtc(dev(W, yb), sch, L)), oracle.dwtc_lifting(yb, sch, L, fw=False)), (shape, "inv")
    x = rng_array((1 << 20,), dtype, 3)
    ye = oracle.dwt_lifting(x, sch, 20)
    xe = oracle.dwt_lifting(ye, sch, 20, fw=False)
    assert np.array_equal(host(W, W.dwt(dev(W, x), sch, 20)), ye)
    assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, 20)), xe)
    W.set_option("WL_LIFT_REGTAIL", 0)
    assert np.array_equal(host(W, W.dwt(dev(W, x), sch, 20)), ye)
    assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, 20)), xe)
    xs = rng_array((2048,), dtype, 9)
    y0 = host(W, W.dwt(dev(W, xs), sch, 11))
    assert W.last_kernel() == "k_tail_lift" and np.array_equal(y0, oracle.dwt_lifting(xs, sch, 11))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lifting_register_tail_2d(gpu, W, oracle, dtype):
    """k_tail_lift2d_reg (every remaining level of a power-of-two block <= 64 x 64 in one wave: a lane holds a whole row /
    column in registers, compile-time wrap and in-bounds / boundary summation forms): every block size from 2 to 64, every
    depth, the three scheme shapes, forward and inverse, in place, as the end of a larger transform -- bit for bit against
    the oracle and against the LDS workgroup tail it replaces."""
    for sname in ("cdf97", "db2", "haar"):
        sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
        n = 2
        while n <= 64:
            x = rng_array((n, n), dtype, n + len(sname))
            Lmax = W.maxtransformlevels(n)
            for L in range(1, Lmax + 1):
                ye = oracle.dwt_lifting(x, sch, L)
                assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L)
                t = dev(W, x)
                W.dwt_(t, sch, L)
                assert np.array_equal(host(W, t), ye), (sname, n, L, "in place")
                xe = oracle.dwt_lifting(ye, sch, L, fw=False)
                assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (sname, n, L, "inv")
                t = dev(W, ye)
                W.idwt_(t, sch, L)
                assert np.array_equal(host(W, t), xe), (sname, n, L, "inv in place")
                with W.options(WL_LIFT_REGTAIL2D=0):
                    assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L, "lds tail")
                    assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (sname, n, L, "lds tail inv")
            n *= 2
        for n, L in ((512, 9), (256, 6), (1024, 5)):
            x = rng_array((n, n), dtype, n)
            ye = oracle.dwt_lifting(x, sch, L)
            assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L)
            assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), oracle.dwt_lifting(ye, sch, L, fw=False)), (sname, n, L, "inv")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lifting_lds_tail_2d_thread_per_line(gpu, W, oracle, dtype):
    """k_tail_lift2d_lds (round 4: every remaining level of a power-of-two block of <= 128 x 128 Float32 / 64 x 64 Float64 in
    one workgroup, a thread per line, 256 threads stage and store): every block size from 2 up (WL_LIFT_LDSTAIL2D_MIN = 2 sends
    the sizes the register tail normally takes here too), every depth, the three scheme shapes, forward and inverse, in place,
    as the end of a larger transform -- bit for bit against the oracle and against the register tail / tile launches."""
    nmax = 128 if dtype == np.float32 else 64
    for sname in ("cdf97", "db2", "haar"):
        sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
        n = 2
        while n <= nmax:
            x = rng_array((n, n), dtype, n + 2 * len(sname))
            for L in range(1, W.maxtransformlevels(n) + 1):
                ye = oracle.dwt_lifting(x, sch, L)
                xe = oracle.dwt_lifting(ye, sch, L, fw=False)
                with W.options(WL_LIFT_LDSTAIL2D_MIN=2, WL_LIFT_LDSTAIL2D_FMIN=2, WL_LIFT_LDSTAIL2D_FMAX=128):
                    assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L)
                    t = dev(W, x)
                    W.dwt_(t, sch, L)
                    assert np.array_equal(host(W, t), ye), (sname, n, L, "in place")
                    assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (sname, n, L, "inv")
                    t = dev(W, ye)
                    W.idwt_(t, sch, L)
                    assert np.array_equal(host(W, t), xe), (sname, n, L, "inv in place")
                with W.options(WL_LIFT_LDSTAIL2D=0):
                    assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L, "register tail / tiles")
                    assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (sname, n, L, "register tail / tiles inv")
            n *= 2
        for n, L in ((512, 9), (256, 2), (1024, 4), (2048, 11)):
            x = rng_array((n, n), dtype, n)
            ye = oracle.dwt_lifting(x, sch, L)
            assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L)
            assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), oracle.dwt_lifting(ye, sch, L, fw=False)), (sname, n, L, "inv")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lifting_lds_tail_3d(gpu, W, oracle, dtype):
    """k_tail_lift3d (round 4: every remaining level of a power-of-two cube of <= 32^3 Float32 / 16^3 Float64 in one workgroup's
    LDS, a thread per line, planes -> rows -> columns and back): every cube from 8 to 64, every depth, the three scheme shapes,
    forward and inverse, in place, as the end of a larger transform -- bit for bit against the oracle and against the per-axis
    launches it replaces."""
    for sname in ("cdf97", "db2", "haar"):
        sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
        for n in (8, 16, 32, 64):
            x = rng_array((n, n, n), dtype, n + len(sname))
            for L in range(1, W.maxtransformlevels(n) + 1):
                ye = oracle.dwt_lifting(x, sch, L)
                assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L)
                t = dev(W, x)
                W.dwt_(t, sch, L)
                assert np.array_equal(host(W, t), ye), (sname, n, L, "in place")
                xe = oracle.dwt_lifting(ye, sch, L, fw=False)
                assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (sname, n, L, "inv")
                t = dev(W, ye)
                W.idwt_(t, sch, L)
                assert np.array_equal(host(W, t), xe), (sname, n, L, "inv in place")
                with W.options(WL_LIFT_TAIL3D=0):
                    assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L, "per-axis launches")
                    assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (sname, n, L, "per-axis launches inv")
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    x = rng_array((128, 128, 128), dtype, 9)
    for L in (7, 4):
        ye = oracle.dwt_lifting(x, sch, L)
        assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), L
        assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), oracle.dwt_lifting(ye, sch, L, fw=False)), (L, "inv")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lifting_any_even_size_tile_kernel(gpu, W, oracle, dtype):
    """k_lift2d_gtile (one 2-D lifting level of ANY even size per launch: one wave per 64 x 64 tile, a lane holds a tile row /
    column in registers, halo = the scheme's dependency cone, in-bounds / boundary summation forms selected by the global
    position): sizes that are not multiples of 8, blocks barely larger than a tile, partial edge tiles, levels below the
    streaming kernels of a larger transform; the three scheme shapes, forward and inverse -- bit for bit against the oracle
    and against the one-thread-per-element kernels it replaces."""
    for sname in ("cdf97", "db2", "haar"):
        sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
        for n, Ls in ((66, (1,)), (100, (1, 2)), (130, (1,)), (250, (1,)), (500, (1, 2)), (1000, (3,)), (132, (2,)), (3000, (3,)), (72, (3,))):
            if n >= 3000 and (sname != "cdf97" or dtype == np.float64):
                continue
            x = rng_array((n, n), dtype, n + len(sname))
            for L in Ls:
                ye = oracle.dwt_lifting(x, sch, L)
                y = host(W, W.dwt(dev(W, x), sch, L))
                assert np.array_equal(y, ye), (sname, n, L, W.last_kernel(), int((y != ye).sum()))
                xe = oracle.dwt_lifting(ye, sch, L, fw=False)
                xr = host(W, W.idwt(dev(W, ye), sch, L))
                assert np.array_equal(xr, xe), (sname, n, L, "inv", W.last_kernel(), int((xr != xe).sum()))
                if n <= 1000:
                    with W.options(WL_LIFT_GTILE=0):
                        assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L, "generic")
                        assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (sname, n, L, "generic inv")
    # the rank-generic driver (3-D volumes of any even size, in-place 2-D level 1): one k_lift_any launch per axis and level
    for sname in ("cdf97", "db2", "haar"):
        sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
        for n, L in ((10, 1), (36, 2), (100, 2), (6, 1)):
            x = rng_array((n, n, n), dtype, n)
            ye = oracle.dwt_lifting(x, sch, L)
            assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L, "3-D")
            xe = oracle.dwt_lifting(ye, sch, L, fw=False)
            assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (sname, n, L, "3-D inv")
            with W.options(WL_LIFT_ANY=0):
                assert np.array_equal(host(W, W.dwt(dev(W, x), sch, L)), ye), (sname, n, L, "3-D generic")
                assert np.array_equal(host(W, W.idwt(dev(W, ye), sch, L)), xe), (sname, n, L, "3-D generic inv")
        for n, L in ((250, 1), (1000, 3), (36, 2)):
            x = rng_array((n, n), dtype, n)
            ye = oracle.dwt_lifting(x, sch, L)
            t = dev(W, x)
            W.dwt_(t, sch, L)
            assert np.array_equal(host(W, t), ye), (sname, n, L, "2-D in place")
            t = dev(W, ye)
            W.idwt_(t, sch, L)
            assert np.array_equal(host(W, t), oracle.dwt_lifting(ye, sch, L, fw=False)), (sname, n, L, "2-D in place inv")
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    x = rng_array((250, 250), dtype, 1)
    W.dwt(dev(W, x), sch, 1)
    assert W.last_kernel() == "k_lift2d_gtile", W.last_kernel()
    W.idwt(dev(W, x), sch, 1)
    assert W.last_kernel() == "k_lift2d_gtile", W.last_kernel()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lifting_lines_any_even_length(gpu, W, oracle, dtype):
    """k_lift1d_gtile (one 1-D lifting level of lines of ANY even length: a lane owns 28 pairs and holds their 32-pair window
    in registers): lengths that are not multiples of 8 or have a large odd factor, alone, below the streaming kernels and
    above the LDS tail, in place, batched columns -- bit for bit against the oracle."""
    for sname in ("cdf97", "db2", "haar"):
        sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
        for n, L in ((1000000, 6), (3000, 3), (4100, 2), (2100, 1), (6 * 4096, 13), (100004, 2), (2052, 2)):
            if n >= 1000000 and sname != "cdf97":
                continue
            x = rng_array((n,), dtype, n % 1000 + len(sname))
            ye = oracle.dwt_lifting(x, sch, L)
            y = host(W, W.dwt(dev(W, x), sch, L))
            assert "generic" not in W.last_kernel(), (sname, n, L, W.last_kernel())
            assert np.array_equal(y, ye), (sname, n, L, W.last_kernel(), int((y != ye).sum()))
            t = dev(W, x)
            W.dwt_(t, sch, L)
            assert np.array_equal(host(W, t), ye), (sname, n, L, "in place")
            xe = oracle.dwt_lifting(ye, sch, L, fw=False)
            xr = host(W, W.idwt(dev(W, ye), sch, L))
            assert "generic" not in W.last_kernel(), (sname, n, L, W.last_kernel())
            assert np.array_equal(xr, xe), (sname, n, L, "inv", int((xr != xe).sum()))
            t = dev(W, ye)
            W.idwt_(t, sch, L)
            assert np.array_equal(host(W, t), xe), (sname, n, L, "inv in place")
        for shape, L in (((3000, 7), 3), ((1028, 40), 2), ((100, 33), 2)):
            xb = rng_array(shape, dtype, shape[1])
            yb = oracle.dwtc_lifting(xb, sch, L)
            assert np.array_equal(host(W, W.dwtc(dev(W, xb), sch, L)), yb), (sname, shape)
            assert np.array_equal(host(W, W.idwtc(dev(W, yb), sch, L)), oracle.dwtc_lifting(yb, sch, L, fw=False)), (sname, shape, "inv")


def test_lifting_equals_filter_on_gpu(gpu, W):
    """test/transforms.jl:57-128 (tolerance 1e-10*sqrt(len)) on the device results."""
    for nd in (1, 2, 3):
        x = rng_array((32,) * nd, np.float64, nd)
        for wname in ("db1", "db2"):
            wf, wl = W.wavelet(getattr(W.WT, wname)), W.wavelet(getattr(W.WT, wname), W.WT.Lifting)
            for L in (5, 0, 1, 2):
                yf, yl = host(W, W.dwt(dev(W, x), wf, L)), host(W, W.dwt(dev(W, x), wl, L))
                assert np.linalg.norm(yf - yl) <= 1e-9 * np.sqrt(x.size)
                assert np.linalg.norm(host(W, W.idwt(dev(W, yl), wl, L)) - x) <= 1e-9 * np.sqrt(x.size)


# ---- dwt! / types / views (test/transforms.jl:130-201) ---------------------------------------------
def test_inplace_api_and_types(gpu, W, oracle):
    import torch
    wt = W.wavelet(W.WT.db2)
    x = rng_array((64,), np.float32, 1)
    xd = dev(W, x)
    y = W.similar(xd)
    r = W.dwt_(y, xd, wt, 2)
    assert r is y and y.dtype == torch.float32
    assert np.array_equal(host(W, y), oracle.dwt_filter(x, wt.qmf, 2))
    assert np.array_equal(host(W, W.dwt_(W.similar(xd), xd, wt)), host(W, W.dwt(xd, wt)))
    # dwt_oop!(y, x, wt, L) for a filter and for a lifting scheme; column-wise into a given y
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    y2 = W.similar(xd)
    assert W.dwt_oop_(y2, xd, sch, 3) is y2 and np.array_equal(host(W, y2), oracle.dwt_lifting(x, sch, 3))
    assert np.array_equal(host(W, xd), x)                                  # x untouched
    assert np.array_equal(host(W, W.idwt_oop_(W.similar(xd), y2, sch, 3)), oracle.dwt_lifting(oracle.dwt_lifting(x, sch, 3), sch, 3, fw=False))
    assert np.array_equal(host(W, W.dwt_oop_(W.similar(xd), xd, wt, 2)), oracle.dwt_filter(x, wt.qmf, 2))
    xm = rng_array((64, 3), np.float32, 9)
    ym = W.similar(dev(W, xm))
    assert W.dwtc_(ym, dev(W, xm), wt, 2) is ym and np.array_equal(host(W, ym), oracle.dwtc_filter(xm, wt.qmf, 2))
    # Int -> Float (transforms_main.jl:188-190)
    xi = torch.arange(-8, 8, device=gpu)
    yi = W.dwt(xi, wt, 2)
    assert yi.dtype == torch.float64
    assert np.array_equal(host(W, yi), oracle.dwt_filter(np.arange(-8, 8).astype(np.float64), wt.qmf, 2))
    # a row-major (non Julia-layout) 2-D tensor is accepted by dwt (copied into Julia layout)
    a = rng_array((16, 32), np.float64, 3)
    yt = W.dwt(torch.from_numpy(a).to(gpu), wt, 2)
    assert np.array_equal(host(W, yt), oracle.dwt_filter(a, wt.qmf, 2))


def test_degenerate_sizes(gpu, W, oracle):
    """length-0 and length-1 arrays: maxtransformlevels is 0, the transform is the identity copy."""
    import torch
    wt = W.wavelet(W.WT.db2)
    e = torch.zeros(0, dtype=torch.float32, device=gpu)
    assert W.dwt(e, wt).shape == (0,) and W.idwt(e, wt).shape == (0,)
    one = torch.tensor([3.5], dtype=torch.float64, device=gpu)
    assert float(W.dwt(one, wt)[0]) == 3.5 and W.maxtransformlevels(one) == 0
    assert float(W.dwt(one, W.wavelet(W.WT.cdf97, W.WT.Lifting))[0]) == 3.5
    two = dev(W, np.array([1.0, 2.0]))
    assert np.array_equal(host(W, W.dwt(two, wt)), oracle.dwt_filter(np.array([1.0, 2.0]), wt.qmf))


def test_argument_contract_on_gpu(gpu, W):
    import torch
    wt = W.wavelet(W.WT.db2)
    x = dev(W, rng_array((24,), np.float64, 1))
    with pytest.raises(W.ArgumentError, match="power of 2"):
        W.dwt(x, wt, 4)
    with pytest.raises(W.ArgumentError, match="positive"):
        W.dwt(x, wt, -1)
    with pytest.raises(W.ArgumentError, match="in array is out array"):
        W.dwt_(x, x, wt, 1)
    with pytest.raises(W.DimensionMismatch):
        W.dwt_(W.similar(x)[:12], x, wt, 1)
    with pytest.raises(W.ArgumentError, match="square/cube"):
        W.dwt(dev(W, rng_array((8, 16), np.float64, 1)), W.wavelet(W.WT.db2, W.WT.Lifting), 1)
    with pytest.raises(TypeError):
        W.dwt(x.to(torch.complex128), wt, 1)
    # status codes straight from the ABI
    lib = W._lib.load()
    h = C.c_void_p()
    assert lib.wl_ctx_create(0, C.byref(h)) == 0
    dims = (C.c_int64 * 3)(24, 1, 1)
    q = (C.c_double * 4)(*wt.qmf)
    y = W.similar(x)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.wl_dwt_filter(h, 1, p(y), p(x), 1, dims, q, 4, 4, 1, None) == -1
    assert lib.wl_dwt_filter(h, 1, p(y), p(x), 1, dims, q, 4, -1, 1, None) == -2
    assert lib.wl_dwt_filter(h, 1, p(x), p(x), 1, dims, q, 4, 1, 1, None) == -3
    assert lib.wl_dwt_filter(h, 1, p(y), p(x), 4, dims, q, 4, 1, 1, None) == -4
    assert lib.wl_dwt_filter(h, 7, p(y), p(x), 1, dims, q, 4, 1, 1, None) == -8
    assert lib.wl_dwt_filter(h, 1, p(y), p(x), 1, dims, q, 1, 1, 1, None) == -9
    assert lib.wl_dwt_filter(h, 1, None, p(x), 1, dims, q, 4, 1, 1, None) == -10
    assert lib.wl_dwt_filter(h, 1, p(y), p(x), 1, dims, q, 4, 1, 1, None) == 0
    assert lib.wl_stream_sync(h, None) == 0
    assert lib.wl_ctx_destroy(h) == 0


def test_caller_buffer_checks(gpu, W):
    """The entry points that take caller buffers refuse mismatched pairs before any pointer reaches the library:
    element type, device, layout, shape and (wpt) rank -- for the filter, lifting and packet calls alike."""
    import torch
    filt, sch = W.wavelet(W.WT.db2), W.wavelet(W.WT.db2, W.WT.Lifting)
    x32 = dev(W, rng_array((64,), np.float32, 1))
    x64 = dev(W, rng_array((64,), np.float64, 1))
    y32 = W.similar(x32)
    for call in (lambda: W.wpt_(y32, x64, filt), lambda: W.iwpt_(y32, x64, filt), lambda: W.dwt_oop_(y32, x64, sch, 2),
                 lambda: W.idwt_oop_(y32, x64, sch, 2), lambda: W.dwt_(y32, x64, filt, 2)):
        with pytest.raises(TypeError):
            call()
    with pytest.raises(W.HIPError):
        W.wpt_(y32, x32.cpu(), filt)
    with pytest.raises(W.HIPError):
        W.dwt_oop_(y32, x32.cpu(), sch, 2)
    with pytest.raises(W.DimensionMismatch):
        W.wpt_(W.similar(x32)[:32], x32, filt)
    strided = torch.empty(128, dtype=torch.float32, device=gpu)[::2]
    with pytest.raises(W.ArgumentError, match="column-major"):
        W.wpt_(strided, x32, filt)
    with pytest.raises(W.ArgumentError, match="column-major"):
        W.dwt_oop_(strided, x32, sch, 2)
    with pytest.raises(W.ArgumentError, match="column-major"):
        W.wpt_(strided, sch)
    m = dev(W, rng_array((8, 8), np.float32, 1))
    with pytest.raises(TypeError, match="vectors only"):
        W.wpt_(W.similar(m), m, filt)
    # the valid calls still work
    assert np.array_equal(host(W, W.wpt_(y32, x32, filt)), host(W, W.wpt(x32, filt)))


def test_calls_leave_the_current_device_alone(gpu, W, oracle):
    """Every ABI call runs on its context's device and restores the caller's current device (two-device boxes only)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    wt = W.wavelet(W.WT.db4)
    x = rng_array((4096,), np.float32, 3)
    torch.cuda.set_device(0)
    x1 = torch.from_numpy(x).to("cuda:1")
    W.reserve_workspace(x1, 5)
    y1 = W.dwt(x1, wt, 5)
    assert torch.cuda.current_device() == 0
    assert y1.device.index == 1 and np.array_equal(y1.cpu().numpy(), oracle.dwt_filter(x, wt.qmf, 5))
    assert abs(W.median(x1) - float(np.median(x))) < 1e-6 and torch.cuda.current_device() == 0


# ---- batched column-wise ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dwtc_bitexact(gpu, W, oracle, dtype):
    for (n, ns) in ((8, 3), (64, 7), (1024, 33), (4096, 64), (1 << 14, 16)):
        x = rng_array((n, ns), dtype, n + ns)
        Lmax = W.maxtransformlevels(n)
        for fname in ("db4", "haar", "db2"):
            wt = W.wavelet(getattr(W.WT, fname))
            for L in sorted({1, Lmax, Lmax // 2}):
                ye = oracle.dwtc_filter(x, wt.qmf, L)
                y = host(W, W.dwtc(dev(W, x), wt, L))
                assert np.array_equal(y, ye), (n, ns, fname, L, W.last_kernel())
                assert np.array_equal(host(W, W.idwtc(dev(W, ye), wt, L)), oracle.dwtc_filter(ye, wt.qmf, L, fw=False))
        sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
        ye = oracle.dwtc_lifting(x, sch, Lmax)
        assert np.array_equal(host(W, W.dwtc(dev(W, x), sch, Lmax)), ye)
        assert np.array_equal(host(W, W.idwtc(dev(W, ye), sch, Lmax)), oracle.dwtc_lifting(ye, sch, Lmax, fw=False))


def test_many_lines_slab_launches(gpu, W, oracle):
    """gridDim.y is capped at 65535, so batches are launched in slabs of lines; WL_SLAB_LINES=5 forces
    several slabs on a small batch (24 and 7 lines)."""
    W.set_option("WL_SLAB_LINES", int(5))
    wt = W.wavelet(W.WT.db4)
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    for n, ns in (((1 << 15), 24), ((1 << 16), 7)):
        x = rng_array((n, ns), np.float32, n + ns)
        for L in (15, 2):
            ye = oracle.dwtc_filter(x, wt.qmf, L)
            assert np.array_equal(host(W, W.dwtc(dev(W, x), wt, L)), ye)
            assert np.array_equal(host(W, W.idwtc(dev(W, ye), wt, L)), oracle.dwtc_filter(ye, wt.qmf, L, fw=False))
            yl = oracle.dwtc_lifting(x, sch, L)
            assert np.array_equal(host(W, W.dwtc(dev(W, x), sch, L)), yl)
            assert np.array_equal(host(W, W.idwtc(dev(W, yl), sch, L)), oracle.dwtc_lifting(yl, sch, L, fw=False))
    # 2-D inverse uses the line kernel for its dim-1 pass
    x2 = rng_array((1024, 512), np.float32, 5)
    y2 = oracle.dwt_filter(x2, wt.qmf, 2)
    assert np.array_equal(host(W, W.idwt(dev(W, y2), wt, 2)), oracle.dwt_filter(y2, wt.qmf, 2, fw=False))


# ---- wavelet packets ---------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_wpt_bitexact(gpu, W, oracle, dtype):
    rs = np.random.default_rng(4)
    for n in (8, 32, 40, 256, 4096, 1 << 16):
        x = rng_array((n,), dtype, n)
        Lmax = W.maxtransformlevels(n)
        trees = [W.maketree(n, L, "full") for L in range(0, Lmax + 1)] + [W.maketree(n, L, "dwt") for L in (1, Lmax)]
        # a random valid tree
        t = np.zeros(2 ** Lmax - 1, dtype=np.uint8)
        t[0] = 1
        for i in range(1, len(t)):
            parent = (i + 1) // 2 - 1
            t[i] = 1 if (t[parent] and rs.random() < 0.6) else 0
        assert W.isvalidtree(np.zeros(n), t)
        trees.append(t)
        for tree in trees:
            for fname in ("db2", "db4"):
                wt = W.wavelet(getattr(W.WT, fname))
                ye = oracle.wpt_filter(x, wt.qmf, tree)
                y = host(W, W.wpt(dev(W, x), wt, tree))
                assert np.array_equal(y, ye), (n, fname, tree.tolist())
                assert np.array_equal(host(W, W.iwpt(dev(W, ye), wt, tree)), oracle.wpt_filter(ye, wt.qmf, tree, fw=False))
            for sname in ("cdf97", "db2"):
                sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
                ye = oracle.wpt_lifting(x, sch, tree)
                assert np.array_equal(host(W, W.wpt(dev(W, x), sch, tree)), ye), (n, sname, tree.tolist())
                assert np.array_equal(host(W, W.iwpt(dev(W, ye), sch, tree)), oracle.wpt_lifting(ye, sch, tree, fw=False))
    # wpt(x, wt, L) == full tree of depth L; invalid tree is an ArgumentError
    x = rng_array((64,), np.float64, 1)
    wt = W.wavelet(W.WT.db2)
    assert np.array_equal(host(W, W.wpt(dev(W, x), wt, 2)), oracle.wpt_filter(x, wt.qmf, W.maketree(64, 2, "full")))
    bad = W.maketree(64, 2, "full")
    bad[0] = 0
    with pytest.raises(W.ArgumentError, match="invalid tree"):
        W.wpt(dev(W, x), wt, bad)


def _random_tree(W, n, depth, rs, p=0.6):
    nn = 2 ** W.maxtransformlevels(n) - 1
    t = np.zeros(nn, dtype=np.uint8)
    t[0] = 1
    for i in range(1, min(nn, 2 ** depth - 1)):
        t[i] = 1 if (t[(i + 1) // 2 - 1] and rs.random() < p) else 0
    return t


def test_wpt_partial_trees_on_the_packet_kernels(gpu, W, oracle):
    """Round 5: partially split trees (`maketree(n, L, :dwt)`, best-basis-like random trees) take the fused packet kernels too --
    the node bits travel as a per-segment split mask, leaves are passed through inside the launch (transforms_filter.jl:325-356,
    util_main.jl:301-344).  Bit-equal to the oracle in both directions, `last_kernel` pins the tier, and at 2^22 a dwt-shaped and a
    random tree of depth 9 run within 1.5 x of the full tree of the same depth."""
    import torch
    rs = np.random.default_rng(21)
    cases = []
    for n, dtype, depth in ((1 << 18, np.float32, 7), (1 << 16, np.float64, 9), (1 << 14, np.float32, 14), (1 << 12, np.float32, 12),
                            (3 << 14, np.float32, 4), (1 << 20, np.float32, 5)):
        trees = [("dwt", W.maketree(n, depth, "dwt")), ("rand", _random_tree(W, n, depth, rs)), ("rand-sparse", _random_tree(W, n, depth, rs, 0.35))]
        cases.append((n, dtype, depth, trees))
    for n, dtype, depth, trees in cases:
        x = rng_array((n,), dtype, n % 977)
        for tag, tree in trees:
            assert W.isvalidtree(np.zeros(n), tree)
            for fname in ("db4", "haar", "sym5"):
                wt = W.wavelet(getattr(W.WT, fname))
                ye = oracle.wpt_filter(x, wt.qmf, tree)
                y = host(W, W.wpt(dev(W, x), wt, tree))
                kf = W.last_kernel()
                assert np.array_equal(y, ye), (n, tag, fname, kf)
                # (a lone fully split depth is a plain 1-D level: the streaming line kernel; never the one-thread-per-output tier)
                assert kf.startswith("k_wpt_fwd") or kf == "k_fwd1d_stream", (n, tag, fname, kf)
                xr = host(W, W.iwpt(dev(W, ye), wt, tree))
                ki = W.last_kernel()
                assert np.array_equal(xr, oracle.wpt_filter(ye, wt.qmf, tree, fw=False)), (n, tag, fname, ki)
                if n & (n - 1) == 0:
                    assert ki.startswith("k_wpt_inv") or ki == "k_inv1d_stream", (n, tag, fname, ki)
    # timing at 2^22, depth 9
    n, depth = 1 << 22, 9
    wt = W.wavelet(W.WT.db4)
    x = torch.randn(n, dtype=torch.float32, device=gpu)
    y = W.similar(x)

    def t_us(tree, inverse=False):
        f = (lambda: W.iwpt_(y, x, wt, tree)) if inverse else (lambda: W.wpt_(y, x, wt, tree))
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 30 * 1e3
    full = W.maketree(n, depth, "full")
    for inverse in (False, True):
        tf = t_us(full, inverse)
        td = t_us(W.maketree(n, depth, "dwt"), inverse)
        tr = t_us(_random_tree(W, n, depth, rs), inverse)
        assert td <= 1.5 * tf + 10 and tr <= 1.5 * tf + 10, (inverse, tf, td, tr)      # (+10 us: the staged copy of the node bits)


def test_wpt_fast_paths_bitexact_and_pinned(gpu, W, oracle):
    """Round 4: fully split depths run on k_wpt_fwd_multi (up to 3 depths per pass over HBM) and k_wpt_fwd_tail / k_wpt_inv_tail
    (every depth of the segments that fit a workgroup in ONE launch), lifting depths on the fused line kernel -- bit-equal to the
    oracle, and `last_kernel` pins that the fast tier actually ran (transforms_filter.jl:301-359, transforms_lifting.jl:283-319)."""
    cases = [
        # n, dtype, depth (None = maxtransformlevels), filters, expected forward kernel, expected inverse kernel
        (1 << 18, np.float32, 6, ("db4", "haar"), "k_wpt_fwd_multi", "k_wpt_inv_multi"),
        (1 << 18, np.float32, None, ("db4",), "k_wpt_fwd_multi", "k_wpt_inv_multi"),
        (1 << 16, np.float32, 5, ("sym5",), "k_wpt_fwd_multi", "k_wpt_inv_multi"),
        (1 << 14, np.float64, None, ("db2", "db4"), "k_wpt_fwd_multi", "k_wpt_inv_multi"),
        (1 << 12, np.float32, None, ("db4", "sym5", "haar"), "k_wpt_fwd_tail", "k_wpt_inv_tail"),
        (1 << 11, np.float64, 11, ("db3",), "k_wpt_fwd_tail", "k_wpt_inv_tail"),
        (256, np.float32, 8, ("db4",), "k_wpt_fwd_tail", "k_wpt_inv_tail"),
        (3 << 14, np.float32, 4, ("db4",), "k_wpt_fwd_multi", None),          # 49152: segments 49152 .. 6144, no power of two
    ]
    for n, dtype, depth, filters, kf, ki in cases:
        x = rng_array((n,), dtype, n % 1000)
        L = W.maxtransformlevels(n) if depth is None else depth
        tree = W.maketree(n, L, "full")
        for fname in filters:
            wt = W.wavelet(getattr(W.WT, fname))
            ye = oracle.wpt_filter(x, wt.qmf, tree)
            y = host(W, W.wpt(dev(W, x), wt, tree))
            assert W.last_kernel() == kf, (n, L, fname, W.last_kernel())
            assert np.array_equal(y, ye), (n, L, fname)
            xr = host(W, W.iwpt(dev(W, ye), wt, tree))
            if ki is not None:
                assert W.last_kernel() == ki, (n, L, fname, W.last_kernel())
            assert np.array_equal(xr, oracle.wpt_filter(ye, wt.qmf, tree, fw=False)), (n, L, fname)
            # the per-depth tier gives the same bits
            W.set_option("WL_WPT_FAST", 0)
            assert np.array_equal(host(W, W.wpt(dev(W, x), wt, tree)), ye)
            W.clear_options()
    # lifting: every fully split depth = one fused lifting level over the segments
    for n, L in ((1 << 16, 6), (1 << 18, 3)):
        x = rng_array((n,), np.float32, 5)
        tree = W.maketree(n, L, "full")
        for sname in ("cdf97", "db2"):
            sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
            ye = oracle.wpt_lifting(x, sch, tree)
            assert np.array_equal(host(W, W.wpt(dev(W, x), sch, tree)), ye), (n, sname)
            assert W.last_kernel().startswith("k_lift1d") or W.last_kernel().startswith("k_tail_lift"), W.last_kernel()
            assert np.array_equal(host(W, W.iwpt(dev(W, ye), sch, tree)), oracle.wpt_lifting(ye, sch, tree, fw=False))
    # a partially split tree: the node bits are staged by the call -- the caller may reuse its buffer at once
    import torch
    n = 1 << 14
    x = rng_array((n,), np.float32, 9)
    rs = np.random.default_rng(12)
    t = np.zeros(n - 1, dtype=np.uint8)
    t[0] = 1
    for i in range(1, len(t)):
        t[i] = 1 if (t[(i + 1) // 2 - 1] and rs.random() < 0.7) else 0
    wt = W.wavelet(W.WT.db4)
    ye = oracle.wpt_filter(x, wt.qmf, t.copy())
    xd = dev(W, x)
    tt = t.copy()
    yd = W.wpt(xd, wt, tt)
    tt[:] = 1                                   # (before any synchronisation)
    torch.cuda.synchronize()
    assert np.array_equal(host(W, yd), ye)


def test_batch_of_images_bitexact(gpu, W, oracle):
    """wl_dwt_filter_batch: B independent images, every level one launch over all of them == B single transforms (the oracle's)"""
    import torch
    for (n0, n1, nb, L, fname, dtype) in ((512, 512, 5, 4, "db4", np.float32), (1024, 512, 3, 9, "sym5", np.float32), (2048, 2048, 2, 11, "db4", np.float32),
                                           (256, 256, 7, 8, "db2", np.float64), (96, 160, 4, 5, "db4", np.float32), (64, 64, 9, 6, "haar", np.float32)):
        wt = W.wavelet(getattr(W.WT, fname))
        xs = [rng_array((n0, n1), dtype, 100 + i) for i in range(nb)]
        xb = torch.stack([W.to_device(a).t().contiguous() for a in xs]).permute(2, 1, 0)      # n0 x n1 x B, column-major
        assert xb.stride() == (1, n0, n0 * n1)
        yb = W.dwt_batch(xb, wt, L)
        for i in range(nb):
            e = oracle.dwt_filter(xs[i], wt.qmf, L)
            assert np.array_equal(yb[:, :, i].cpu().numpy(), e), (n0, n1, L, fname, i)
        xr = W.idwt_batch(yb, wt, L)
        for i in range(nb):
            e = oracle.dwt_filter(oracle.dwt_filter(xs[i], wt.qmf, L), wt.qmf, L, fw=False)
            assert np.array_equal(xr[:, :, i].cpu().numpy(), e), (n0, n1, L, fname, i)


def test_batch_of_images_padded_image_stride(gpu, W, oracle):
    """wl_dwt_filter_batch through ctypes with image_stride > dims[0] * dims[1] (the documented C ABI; the Python / Julia wrappers
    always pass dense strides): forward and inverse, the plane-batch kernels (n0 >= 256: k_fwd2d_* over blockIdx.y, k_inv2d_stream /
    k_inv2d_lds_long planes) and the per-image fallback; the padding between the images is never written (round-4 advisor finding:
    the inverse plane branch wrote image i at i * n0 * n1 instead of i * image_stride)."""
    import ctypes as C
    import torch
    lib = W._lib.load()
    for (n0, n1, nb, L, fname, dtype, pad) in ((512, 512, 3, 4, "db4", np.float32, 64), (256, 128, 4, 3, "sym5", np.float32, 4096),
                                                (512, 256, 3, 2, "db8", np.float32, 16), (256, 256, 3, 5, "db2", np.float64, 6),
                                                (64, 96, 5, 3, "db4", np.float32, 10), (1024, 512, 2, 6, "db4", np.float32, 20)):
        wt = W.wavelet(getattr(W.WT, fname))
        td = torch.float32 if dtype == np.float32 else torch.float64
        stride = n0 * n1 + pad
        xs = [rng_array((n0, n1), dtype, 700 + i) for i in range(nb)]
        for fw in (1, 0):
            ins = xs if fw else [oracle.dwt_filter(a, wt.qmf, L) for a in xs]
            xb = torch.full((nb * stride,), 7.0, dtype=td, device=gpu)
            yb = torch.full((nb * stride,), -3.0, dtype=td, device=gpu)
            for i, a in enumerate(ins):
                xb[i * stride:i * stride + n0 * n1].copy_(torch.from_numpy(np.ascontiguousarray(a.T).ravel()))
            h, st = W.transforms._context(gpu)
            q = np.ascontiguousarray(wt.qmf, dtype=np.float64)
            rc = lib.wl_dwt_filter_batch(h, 0 if dtype == np.float32 else 1, C.c_void_p(yb.data_ptr()), C.c_void_p(xb.data_ptr()),
                                         (C.c_int64 * 2)(n0, n1), nb, stride, q.ctypes.data_as(C.POINTER(C.c_double)), len(q), L, fw, st)
            assert rc == 0, (rc, n0, n1, fname)
            torch.cuda.synchronize()
            yh = yb.cpu().numpy()
            for i, a in enumerate(ins):
                got = yh[i * stride:i * stride + n0 * n1].reshape(n1, n0).T
                exp = oracle.dwt_filter(a, wt.qmf, L, fw=bool(fw))
                assert np.array_equal(got, exp), (n0, n1, fname, "fw" if fw else "inv", i, W.last_kernel())
                assert np.all(yh[i * stride + n0 * n1:(i + 1) * stride] == -3.0), ("padding written", n0, n1, fname, fw, i)


def test_batched_planes_lds_exchange_inverse(gpu, W, oracle):
    """k_inv2d_lds_long over blockIdx.y (round 4): batches of images (wl_dwt_filter_batch inverse) and the planes of a 3-D level --
    10 taps in Float32, 8 ... 20 taps in Float64, 12 ... 20 in Float32; the first planes of a 3-D level take their approximation
    quadrant from the deeper reconstruction, the others from the coefficient array; bit for bit against the oracle."""
    import torch
    W.set_option("WL_INVLONG2D_MIN_ROWS", 256)
    W.set_option("WL_INVLONG_SHORT_MIN", 0)
    for (n0, n1, nb, L, fname, dtype) in ((512, 512, 5, 3, "sym5", np.float32), (1024, 256, 3, 2, "db8", np.float32), (256, 96, 7, 1, "db10", np.float32),
                                           (512, 256, 4, 2, "db4", np.float64), (256, 512, 3, 2, "db6", np.float64), (768, 64, 2, 1, "coif6", np.float64)):
        wt = W.wavelet(getattr(W.WT, fname))
        ys = [rng_array((n0, n1), dtype, 300 + i) for i in range(nb)]
        yb = torch.stack([W.to_device(a).t().contiguous() for a in ys]).permute(2, 1, 0)      # n0 x n1 x B, column-major
        xr = W.idwt_batch(yb, wt, L)
        assert W.last_kernel() == "k_inv2d_lds_long", (n0, n1, fname, W.last_kernel())
        for i in range(nb):
            assert np.array_equal(xr[:, :, i].cpu().numpy(), oracle.dwt_filter(ys[i], wt.qmf, L, fw=False)), (n0, n1, L, fname, i)
    # 3-D: the dim-1 / dim-2 reconstruction of every plane in one launch, then the axis-3 pass
    for (shape, L, fname, dtype) in (((512, 256, 32), 2, "sym5", np.float32), ((256, 256, 64), 1, "db4", np.float64)):
        wt = W.wavelet(getattr(W.WT, fname))
        y = rng_array(shape, dtype, 17)
        xr = host(W, W.idwt(dev(W, y), wt, L))
        assert np.array_equal(xr, oracle.dwt_filter(y, wt.qmf, L, fw=False)), (shape, fname)


# ---- BASELINE.json full sizes: size-independent properties -------------------------------------------
def test_full_size_properties(gpu, W, oracle):
    """configs[1..3] at full size: round trip, linearity, energy, and exact equality with the oracle on
    sub-problems the oracle finishes in seconds (a column band of the first level / the deep levels)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(42)
    wt = W.wavelet(W.WT.db4)
    # C3: 8192 x 8192 f32, L = 13 (API default), L = 4 and L = 1
    x = torch.randn(8192, 8192, generator=g, dtype=torch.float32).to(gpu).t()     # Julia layout view
    assert W.is_julia_layout(x)
    nx2 = float(torch.linalg.vector_norm(x.double()))
    for L in (13, 4, 1):
        y = W.dwt(x, wt, L)
        assert abs(float(torch.linalg.vector_norm(y.double())) - nx2) / nx2 < 1e-5      # orthogonal: energy
        xr = W.idwt(y, wt, L)
        rel = float(torch.linalg.vector_norm((xr - x).double())) / nx2
        assert rel < 1e-5, (L, rel)                                                     # round trip
    # linearity: dwt(a x + b z) ~ a dwt(x) + b dwt(z)
    z = torch.randn(8192, 8192, generator=g, dtype=torch.float32).to(gpu).t()
    lhs = W.dwt(2.0 * x - 0.5 * z, wt, 13)
    rhs = 2.0 * W.dwt(x, wt, 13) - 0.5 * W.dwt(z, wt, 13)
    assert float(torch.linalg.vector_norm((lhs - rhs).double())) / nx2 < 1e-5
    # exact vs oracle on the coarse part: the level-4.. transform of the LL4 band (512 x 512) equals
    # the oracle run on that band
    y4 = W.dwt(x, wt, 4)
    ll4 = W.to_host(y4[:512, :512])
    y13 = W.to_host(W.dwt(x, wt, 13)[:512, :512])
    assert np.array_equal(y13, oracle.dwt_filter(np.ascontiguousarray(ll4), wt.qmf, 9))
    del x, z, lhs, rhs, y4
    # C2: 1-D db4 f32 2^24, L = 24 -- exact vs the oracle (the 1-D oracle does 2^24 in about a second)
    v = torch.randn(1 << 24, generator=g, dtype=torch.float32)
    yv = W.dwt(v.to(gpu), wt)
    assert np.array_equal(W.to_host(yv), oracle.dwt_filter(v.numpy(), wt.qmf))
    assert float(torch.linalg.vector_norm((W.idwt(yv, wt).cpu() - v).double())) / float(torch.linalg.vector_norm(v.double())) < 1e-5
    # C4: 1-D cdf9/7 lifting f32 2^24, L = 24 -- exact vs the oracle
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    yl = W.dwt(v.to(gpu), sch)
    assert np.array_equal(W.to_host(yl), oracle.dwt_lifting(v.numpy(), sch))
    # C1: 1-D db2 f64 2^20 dwt + idwt
    u = torch.rand(1 << 20, generator=g, dtype=torch.float64)
    w2 = W.wavelet(W.WT.db2)
    yu = W.dwt(u.to(gpu), w2)
    assert np.array_equal(W.to_host(yu), oracle.dwt_filter(u.numpy(), w2.qmf))
    assert float(torch.linalg.vector_norm(W.idwt(yu, w2).cpu() - u)) / float(torch.linalg.vector_norm(u)) < 1e-12


def test_full_size_elementwise_vs_oracle(gpu, W, oracle):
    """The headline configuration compared ELEMENT BY ELEMENT with the oracle at its production size (strip / chunk
    counts, XCD remap, helper waves, reverse walk of the deeper levels as they run in the bench): C3 8192 x 8192 f32 db4
    forward at L = 13, 2, 1 and the inverse at L = 13, 1; 2-D cdf9/7 lifting 4096 x 4096; a C5 shard (8192 signals x
    2^16) on 80 columns incl. the slab and strip boundaries.  The oracle legs use its line-parallel variant, which
    tests/test_oracle_properties.py proves bit-identical to the reference-order single-thread loop."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(42)
    wt = W.wavelet(W.WT.db4)
    xh = torch.randn(8192, 8192, generator=g, dtype=torch.float32).numpy().T          # (8192, 8192) Fortran-ordered view
    xh = np.asfortranarray(xh)
    x = W.to_device(xh)
    W.destroy_contexts()                                   # fresh context: what does this configuration make it hold?
    for L in (13, 2, 1):
        ye = oracle.dwt2d_filter_mt(xh, wt.qmf, L)
        y = W.to_host(W.dwt(x, wt, L))
        assert np.array_equal(y, ye), (L, int((y != ye).sum()))
        if L != 2:
            xr = W.to_host(W.idwt(W.to_device(ye), wt, L))
            xe = oracle.dwt2d_filter_mt(ye, wt.qmf, L, fw=False)
            assert np.array_equal(xr, xe), ("inverse", L, int((xr != xe).sum()))
    # the fast filter-bank path keeps two approximation buffers of N/4 elements: 128 MiB for C3 (round 1: 1 GiB)
    assert W.workspace_held() <= 129 * 2 ** 20, W.workspace_held()
    yb = W.dwt(x, W.wavelet(W.WT.batt2), 2)                # 23-tap filter: two-pass family, grows to the full workspace
    assert W.last_kernel() == "k_vl_lines" and W.workspace_held() > 3 * xh.nbytes
    assert np.array_equal(W.to_host(yb), oracle.dwt2d_filter_mt(xh, W.wavelet(W.WT.batt2).qmf, 2))
    del x, yb
    W.destroy_contexts()
    # 2-D cdf9/7 lifting, 4096 x 4096, full depth
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    xl = np.asfortranarray(torch.randn(4096, 4096, generator=g, dtype=torch.float32).numpy().T)
    yl = W.to_host(W.dwt(W.to_device(xl), sch))
    assert np.array_equal(yl, oracle.dwt_lifting(xl, sch)), "2-D cdf9/7 4096^2"
    # C5 shard: 8192 signals x 2^16, L = 16; oracle on 80 of the columns
    n, ncol = 1 << 16, 8192
    gd = torch.Generator(device=gpu).manual_seed(7)
    xc = torch.randn(ncol, n, generator=gd, dtype=torch.float32, device=gpu).t()
    yc = W.dwtc(xc, wt, 16)
    rs = np.random.default_rng(5)
    cols = sorted(set([0, 1, 63, 64, 255, 256, 4095, 4096, 8190, 8191] + [int(c) for c in rs.integers(0, ncol, 70)]))
    xs = np.asfortranarray(xc[:, cols].cpu().numpy())
    ys = yc[:, cols].cpu().numpy()
    assert np.array_equal(ys, oracle.dwtc_filter(xs, wt.qmf, 16)), "C5 shard columns"


def test_differential_fuzz(gpu, W, oracle):
    """150 random (shape, wavelet, depth, element type, entry point) cases of tests/fuzz_parity.py, bit for bit."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    r = np.random.default_rng(20260926)
    kernels = set()
    for _ in range(150):
        kernels.update(fz.one_case(r))
    assert len(kernels) >= 10, kernels          # the draw must actually spread over the kernel families


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_unaligned_views(gpu, W, oracle, dtype):
    """Arrays that start 4/8 bytes into an allocation (views): the 16-byte fast paths must step aside, results stay
    bit-identical (filter, lifting, batched columns, modwt, threshold!)."""
    import torch
    td = torch.float32 if dtype == np.float32 else torch.float64
    db4, sch = W.wavelet(W.WT.db4), W.wavelet(W.WT.cdf97, W.WT.Lifting)
    for n, L in ((4096, 12), (1 << 16, 5)):
        x = rng_array((n,), dtype, n)
        buf = torch.empty(n + 3, dtype=td, device=gpu)
        xv = buf[1:1 + n]
        xv.copy_(torch.from_numpy(x))
        assert xv.data_ptr() % 16 != 0
        out = torch.empty(n + 3, dtype=td, device=gpu)[1:1 + n]
        W.dwt_(out, xv, db4, L)
        assert np.array_equal(host(W, out), oracle.dwt_filter(x, db4.qmf, L))
        W.idwt_(out, xv, db4, L)
        assert np.array_equal(host(W, out), oracle.dwt_filter(x, db4.qmf, L, fw=False))
        assert np.array_equal(host(W, W.dwt(xv, sch, L)), oracle.dwt_lifting(x, sch, L))
        t = buf[1:1 + n].clone()
        assert np.array_equal(host(W, W.modwt(xv, db4, 3)), oracle.modwt(x, db4.qmf, 3))
        # packet transforms: the 16-byte kernels (k_wpt_fwd_multi / _inv_multi / _tail) must step aside too
        for tree in (W.maketree(n, min(L, 6), "full"), min(L, 5)):
            wo = torch.empty(n + 3, dtype=td, device=gpu)[1:1 + n]
            W.wpt_(wo, xv, db4, tree)
            assert not W.last_kernel().startswith("k_wpt"), W.last_kernel()
            tr = tree if not isinstance(tree, int) else W.maketree(n, tree, "full")
            assert np.array_equal(host(W, wo), oracle.wpt_filter(x, db4.qmf, tr))
            wi = torch.empty(n + 3, dtype=td, device=gpu)[1:1 + n]
            W.iwpt_(wi, wo, db4, tree)
            assert np.array_equal(host(W, wi), oracle.wpt_filter(host(W, wo), db4.qmf, tr, fw=False))
            wl = xv.clone() if False else torch.empty(n + 3, dtype=td, device=gpu)[1:1 + n]
            wl.copy_(xv)
            W.wpt_(wl, sch, tree)
            assert np.array_equal(host(W, wl), oracle.wpt_lifting(x, sch, tr))
        tv = torch.empty(n + 3, dtype=td, device=gpu)[1:1 + n]
        tv.copy_(xv)
        W.threshold_(tv, W.SoftTH(), 0.3)
        assert np.array_equal(host(W, tv), oracle.threshold(x, "soft", 0.3))
        W.dwt_(tv, sch, L)                                    # in place on the view
    # 2-D view with an odd leading offset
    a = rng_array((512, 64), dtype, 9)
    big = torch.empty(512 * 64 + 5, dtype=td, device=gpu)
    av = big[1:1 + 512 * 64].view(64, 512).t()               # Julia layout, data pointer 4/8 bytes off
    av.copy_(torch.from_numpy(a))
    assert W.is_julia_layout(av) and av.data_ptr() % 16 != 0
    assert np.array_equal(host(W, W.dwt(av, db4, 3)), oracle.dwt_filter(a, db4.qmf, 3))
    assert np.array_equal(host(W, W.dwtc(av, db4, 4)), oracle.dwtc_filter(a, db4.qmf, 4))


def test_whole_c5_batch_on_one_gpu(gpu, W):
    """BASELINE config C5 in full (65536 signals x 2^16 samples = 2^32 Float32 elements, 16 GiB) on ONE GPU: 64-bit
    indexing beyond 2^32 elements, slab launches (gridDim.y <= 65535), per-column agreement with the 1-D transform
    (bit-exact, columns on both sides of the slab boundary) and the round trip.  x, y and the reconstruction are 16 GiB
    each; the library's workspace for it is the approximation ping-pong only: 16 GiB (it was 64 GiB)."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 72 * 2 ** 30:
        pytest.skip("needs ~66 GB of free HBM")
    wt = W.wavelet(W.WT.db4)
    n, ns = 1 << 16, 65536
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.empty(ns, n, dtype=torch.float32, device=gpu).normal_(generator=g).t()
    assert x.numel() == 2 ** 32
    W.destroy_contexts()
    y = W.dwtc(x, wt, 16)
    assert W.workspace_held() <= 2 ** 34 + 4096, W.workspace_held()
    for j in (0, 1, 32767, 32768, 65534, 65535):
        assert torch.equal(W.dwt(x[:, j].contiguous(), wt, 16), y[:, j]), j
    # round 4: EVERY column, not a sample -- the 1-D transform of each of the 65536 signals (a different kernel chain: one line
    # instead of a batch; itself pinned to the oracle at this length by test_full_size_properties) must give the bits of the
    # batch's column.  Mismatches are counted on the device; ~25 us per column.
    col = torch.empty(n, dtype=torch.float32, device=gpu)
    yc = torch.empty(n, dtype=torch.float32, device=gpu)
    bad = torch.zeros((), dtype=torch.int64, device=gpu)
    W.reserve_workspace(col, 16)
    for j in range(ns):
        col.copy_(x[:, j])
        W.dwt_oop_(yc, col, wt, 16)
        bad += (yc != y[:, j]).any()
    assert int(bad.item()) == 0, "%d of %d columns of the batch differ from their 1-D transform" % (int(bad.item()), ns)
    xr = W.idwtc(y, wt, 16)
    assert (xr - x).abs().max().item() < 1e-4
    del x, y, xr
    torch.cuda.empty_cache()


def test_2d_array_beyond_2_31_elements(gpu, W, oracle):
    """A 2-D transform whose element offsets do not fit 32 bits: 65536 x 49152 Float32 (3.2e9 elements, 12.9 GB; the headline
    kernels index with int64_t -- C5's 2^32 elements only cover dwtc).  Level 1 is compared bit for bit with the reference order
    on sampled columns whose offsets lie beyond 2^31 and 2^32 elements (dim-2 pass restated in numpy, un-fused, ascending /
    descending tap order as wl_internal.h states it; dim-1 pass = the oracle's 1-D transform of that column); the deeper levels
    through the exact sub-problem (the L = 14 result on the LL4 block equals the oracle run on the L = 4 result's LL4 block);
    the whole array through the round trip."""
    import torch
    n0, n1 = 65536, 49152
    assert n0 * n1 > 2 ** 31
    g = torch.Generator(device=gpu).manual_seed(77)
    x = torch.randn(n1, n0, generator=g, dtype=torch.float32, device=gpu).t()          # Julia layout, n0 x n1
    assert W.is_julia_layout(x)
    wt = W.wavelet(W.WT.db4)
    h = wt.qmf.astype(np.float32)
    F = len(h)
    gg = np.array([h[m] if m % 2 == 0 else np.float32(h[m] * np.float32(-1)) for m in range(F)], dtype=np.float32)
    y = W.dwt(x, wt, 2)
    assert W.last_kernel() == "k_fwd2d_pair", W.last_kernel()
    nx = n1 // 2

    def col(j):
        return x[:, j % n1].cpu().numpy()

    for k in (0, 3, nx // 2 + 5, 16384 + 7, nx - 1, nx - 2):         # output column k: input columns 2k .. 2k+7 (periodic)
        s = h[0] * col(2 * k)
        for m in range(1, F):
            s = s + h[m] * col(2 * k + m)
        d = gg[F - 1] * col(2 * k + 2 - F)
        for m in range(F - 2, -1, -1):
            d = d + gg[m] * col(2 * k + 1 - m)
        es = oracle.dwt_filter(s, wt.qmf, 1)
        ed = oracle.dwt_filter(d, wt.qmf, 1)
        # scaling column k: its detail rows (the DS quadrant) are final after level 1; detail column k: the whole column is
        assert np.array_equal(y[n0 // 2:, k].cpu().numpy(), es[n0 // 2:]), ("ds", k)
        assert np.array_equal(y[:, nx + k].cpu().numpy(), ed), ("sd/dd", k)
    # deeper levels: exact sub-problem
    y4 = W.dwt(x, wt, 4)
    ll4 = W.to_host(y4[:n0 >> 4, :n1 >> 4])
    del y4
    yL = W.dwt(x, wt, 14)
    assert np.array_equal(W.to_host(yL[:n0 >> 4, :n1 >> 4]), oracle.dwt2d_filter_mt(np.asfortranarray(ll4), wt.qmf, 10))
    # level-1 quadrants of the deep transform equal those of the L = 2 call (they are final after level 1)
    assert torch.equal(yL[:, nx:], y[:, nx:]) and torch.equal(yL[n0 // 2:, :nx], y[n0 // 2:, :nx])
    del y
    xr = W.idwt(yL, wt, 14)
    del yL
    num = float(torch.linalg.vector_norm((xr - x).double()))
    den = float(torch.linalg.vector_norm(x.double()))
    assert num / den < 1e-5, num / den
    del xr, x
    W.destroy_contexts()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_3d_forward_level_in_slabs(gpu, W, oracle, dtype):
    """Round 5 experiment kept as an option (WL_3D_SLAB, off by default: measured slower, wl_axis.hip): the forward 3-D level in SLABS of
    output plane pairs (axis-3 pass of a slab, then the fused 2-D level kernel on exactly the planes it produced).  Forced here on
    small boxes (WL_3D_SLAB_MIN_MIB = 0, slabs of 8 / 16 pairs): bit for bit against the whole-box passes and the oracle, including
    the slabs whose detail planes wrap around the end of the detail half (transforms_filter.jl:246-290)."""
    for shape, L in (((256, 64, 64), 2), ((256, 32, 128), 1), ((512, 16, 96), 1)):
        x = rng_array(shape, dtype, shape[2])
        xd = dev(W, x)
        for fname in ("db4", "haar", "db2", "db5"):
            wt = W.wavelet(getattr(W.WT, fname))
            W.set_option("WL_3D_SLAB", 0)
            W.set_option("WL_3D_ONE", 0)                                # (round 6: the one-pass level would take these boxes)
            y0 = host(W, W.dwt(xd, wt, L))
            assert W.last_kernel() == "k_fwd_axis_stream"
            for slab in (8, 16):
                W.set_option("WL_3D_SLAB", slab)
                W.set_option("WL_3D_SLAB_MIN_MIB", 0)
                y1 = host(W, W.dwt(xd, wt, L))
                assert np.array_equal(y0, y1), (shape, fname, slab, int((y0 != y1).sum()))
            W.clear_options()
            if shape[0] * shape[1] * shape[2] <= 1 << 20:
                assert np.array_equal(y0, oracle.dwt_filter(x, wt.qmf, L)), (shape, fname)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_3d_one_pass_level(gpu, W, oracle, dtype):
    """Round 6: one forward 3-D level in ONE pass over HBM (k_fwd3d_one, wl_fwd3d.hip: whole dim-1 lines per workgroup, tiles of 4 raw
    planes along dim 3, a march along dim 2) instead of the axis-3 pass + the plane kernel.  Bit for bit against the oracle and
    against the two-pass tier: both element types, every filter length it takes (2 .. 8 taps), every line length (32 .. 1024 rows in multiples
    of 8 / 4: 1 .. 8 waves per workgroup, 16- and 8-byte lanes, partly filled last waves), the tile whose raw planes wrap around the end of dim 3, segments of 8 .. 64 columns,
    with and without a deeper level behind it (approximation corner to the ping-pong buffer or into y)
    (transforms_filter.jl:246-263)."""
    for shape, L in (((256, 16, 16), 1), ((128, 32, 16), 1), ((256, 32, 48), 2), ((512, 16, 20), 1), ((1024, 16, 16), 2), ((256, 64, 32), 3),
                     ((512, 64, 16), 1), ((128, 64, 64), 2), ((200, 24, 20), 1), ((240, 40, 16), 2), ((320, 16, 16), 1), ((72, 16, 16), 1),
                     ((1000, 16, 16), 1), ((136, 48, 24), 1), ((300, 16, 16), 1), ((180, 24, 20), 1), ((900, 16, 16), 1),
                     ((256, 20, 18), 1), ((200, 30, 22), 1), ((304, 36, 28), 2), ((128, 70, 26), 1)):         # (from 200 x 24 x 20 on: lines that do not fill the last wave, 8-byte lanes on two to eight waves, dim-2 / dim-3 extents
                                                                       #  that are not multiples of 8 / 4: the last segment / tile overlaps its neighbour)
        x = rng_array(shape, dtype, shape[1] + shape[2])
        xd = dev(W, x)
        for fname in ("db4", "haar", "db2", "db3", "db5"):
            wt = W.wavelet(getattr(W.WT, fname))
            W.set_option("WL_3D_ONE", 0)
            y0 = host(W, W.dwt(xd, wt, L))
            k0 = W.last_kernel()
            assert k0 != "k_fwd3d_one", k0
            W.clear_options()
            if fname == "db4":
                assert np.array_equal(y0, oracle.dwt_filter(x, wt.qmf, L)), (shape, fname)
            for tj in (64, 8, 16, 32):
                W.set_option("WL_3D_ONE_MIN", 0)
                W.set_option("WL_3D_ONE_WAVES", 0)                      # (keep the requested segment length on these small boxes)
                W.set_option("WL_3D_ONE_TJ", tj)
                y1 = host(W, W.dwt(xd, wt, L))
                k = W.last_kernel()
                W.clear_options()
                assert k == ("k_fwd3d_one" if (len(wt.qmf) <= 8 or dtype == np.float32) else k0), (shape, fname, k)     # (10 taps: Float32, 8-byte lanes)
                assert np.array_equal(y0, y1), (shape, L, fname, tj, int((y0 != y1).sum()))
    # the default gate: 128^3 and up take it without options
    for n in (128, 256):
        x = rng_array((n, n, n), dtype, 77)
        xd = dev(W, x)
        wt = W.wavelet(W.WT.db4)
        y1 = host(W, W.dwt(xd, wt, 2))
        assert W.last_kernel() == "k_fwd3d_one", W.last_kernel()
        W.set_option("WL_3D_ONE", 0)
        y0 = host(W, W.dwt(xd, wt, 2))
        W.clear_options()
        assert np.array_equal(y0, y1), (n, int((y0 != y1).sum()))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_3d_inverse_one_pass_level(gpu, W, oracle, dtype):
    """Round 6: one INVERSE 3-D level in one pass over HBM (k_inv3d_one, wl_inv3d.hip: whole dim-1 lines per workgroup, tiles of two
    output column pairs along dim 2, a march along dim 3 with a ring of reconstructed planes) instead of the plane kernel + the axis-3
    pass.  Bit for bit against the oracle and the two-pass tier: both element types, 2 .. 8 taps, lines of 32 .. 1024 rows (16- and
    8-byte lanes, partly filled waves), dim-2 / dim-3 extents that are not multiples of 4 / of the segment, segments of every
    length, with the approximation octant in the coefficient array (L = 1) and in the ping-pong buffer (L > 1)
    (transforms_filter.jl:264-287)."""
    for shape, L in (((256, 16, 16), 1), ((128, 32, 16), 1), ((256, 32, 48), 2), ((512, 16, 20), 1), ((1024, 16, 16), 2), ((256, 64, 32), 3),
                     ((512, 64, 16), 1), ((128, 64, 64), 2), ((200, 24, 20), 1), ((240, 40, 16), 2), ((72, 16, 16), 1), ((1000, 16, 16), 1),
                     ((300, 16, 16), 1), ((256, 20, 18), 1), ((200, 30, 22), 1), ((128, 70, 26), 1)):
        if dtype == np.float64 and shape[0] > 512:
            continue
        x = rng_array(shape, dtype, shape[1] + shape[2] + 1)
        for fname in ("db4", "haar", "db2", "db3", "db5"):
            wt = W.wavelet(getattr(W.WT, fname))
            yd = W.dwt(dev(W, x), wt, L)
            W.set_option("WL_I3D_ONE", 0)
            x0 = host(W, W.idwt(yd, wt, L))
            k0 = W.last_kernel()
            assert k0 != "k_inv3d_one", k0
            W.clear_options()
            if fname == "db4":
                assert np.array_equal(x0, oracle.dwt_filter(host(W, yd), wt.qmf, L, fw=False)), (shape, fname)
            for tk in (32, 4, 8, 16):
                for key in ("WL_I3D_ONE_MIN", "WL_I3D_ONE_MIN_LONG", "WL_I3D_ONE_MIN_ANY", "WL_I3D_ONE_WAVES"):
                    W.set_option(key, 0)
                W.set_option("WL_I3D_ONE_F64_FMAX", 8)
                W.set_option("WL_I3D_ONE_TK", tk)
                x1 = host(W, W.idwt(yd, wt, L))
                k = W.last_kernel()
                W.clear_options()
                assert k == ("k_inv3d_one" if len(wt.qmf) <= 8 else k0), (shape, fname, k)
                assert np.array_equal(x0, x1), (shape, L, fname, tk, int((x0 != x1).sum()))
    for n in (128, 256):                               # the default gate: 2^20 elements for 2 / 4 taps (2^27 for 6 / 8 taps, Float32 only)
        x = rng_array((n, n, n), dtype, 78)
        wt = W.wavelet(W.WT.db2)
        yd = W.dwt(dev(W, x), wt, 2)
        x1 = host(W, W.idwt(yd, wt, 2))
        assert W.last_kernel() == "k_inv3d_one", W.last_kernel()
        W.set_option("WL_I3D_ONE", 0)
        x0 = host(W, W.idwt(yd, wt, 2))
        W.clear_options()
        assert np.array_equal(x0, x1), (n, int((x0 != x1).sum()))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_3d_small_levels_in_one_launch(gpu, W, oracle, dtype):
    """Round 6: the 3-D levels between the streaming sizes and the one-workgroup tail (4096 < elements <= 2^18) in ONE launch each,
    forward and inverse (k_level3_lds, wl_level3.hip: LDS blocks of 4^3 / 8^3 coefficient pairs with recomputed halos) instead of
    three single-axis launches.  Bit for bit against the oracle and against the three-launch tier: both element types, 2 .. 10
    taps, cubes and non-cubic boxes (extents that are multiples of 8 but not of 16 take the 4^3 blocks), both block sizes forced,
    levels whose windows wrap more than once around a short axis (transforms_filter.jl:246-287)."""
    for shape, L in (((32, 32, 32), 1), ((64, 64, 64), 2), ((64, 32, 16), 1), ((48, 40, 24), 1), ((16, 16, 32), 1), ((32, 64, 128), 3), ((64, 64, 64), 6),
                     ((96, 96, 96), 2), ((40, 56, 72), 1), ((50, 50, 50), 1), ((36, 20, 28), 1), ((100, 60, 36), 2)):          # (from 96^3 on: not shapes of the axis kernels, the gate is 2^20 there; the last three: extents that are not
                                                                                  #  multiples of 8 -- the last block overlaps its neighbour)
        x = rng_array(shape, dtype, shape[0] + shape[2])
        xd = dev(W, x)
        for fname in ("db4", "haar", "db2", "db3", "db5"):
            wt = W.wavelet(getattr(W.WT, fname))
            ye = oracle.dwt_filter(x, wt.qmf, L)
            for opts in ({}, {"WL_LEVEL3_P8_MIN": 1}):
                W.set_option("WL_3D_ONE_MIN_ANY", 1 << 40)              # (96^3 would otherwise take the one-pass levels)
                W.set_option("WL_I3D_ONE_MIN_ANY", 1 << 40)
                for k, v in opts.items():
                    W.set_option(k, v)
                y = host(W, W.dwt(xd, wt, L))
                kf = W.last_kernel()
                xr = host(W, W.idwt(dev(W, ye), wt, L))
                ki = W.last_kernel()
                W.clear_options()
                assert kf == ki == "k_level3_lds", (shape, fname, kf, ki)
                assert np.array_equal(y, ye), (shape, L, fname, opts, int((y != ye).sum()))
                assert np.array_equal(xr, oracle.dwt_filter(ye, wt.qmf, L, fw=False)), (shape, L, fname, opts)
            W.set_option("WL_LEVEL3", 0)
            y0 = host(W, W.dwt(xd, wt, L))
            assert W.last_kernel() != "k_level3_lds"
            W.clear_options()
            assert np.array_equal(y0, ye), (shape, fname)


def test_long_filter_tiles_bitexact(gpu, W, oracle):
    """Round 5: the cache-resident levels (128 .. 1024 rows; WL_TILE_LONG_MAX) of the 12..20-tap filters take the LDS tile kernel, two
    levels per launch (wl_tile.hip: the dim-1 window generalised to 24 / 32 rows, detail rows shifted by 8 / 12) instead of one streaming
    launch per level and two line passes at 128^2.  Bit for bit against the oracle and against the per-level tier, every filter
    length, square and non-square blocks, odd numbers of remaining levels (transforms_filter.jl:113-188)."""
    for (n0, n1, L) in ((1024, 1024, 10), (512, 512, 3), (256, 256, 8), (128, 128, 7), (1024, 256, 2), (128, 512, 3), (256, 1024, 8), (2048, 2048, 4)):
        x = rng_array((n0, n1), np.float32, n0 + n1)
        xd = dev(W, x)
        for fname in ("db6", "db7", "sym8", "db9", "db10", "coif4", "beyl"):
            wt = W.wavelet(getattr(W.WT, fname))
            if n0 * n1 > 1 << 20 and fname not in ("sym8", "db10"):
                continue
            y = host(W, W.dwt(xd, wt, L))
            if max(n0, n1) <= 1024:
                assert W.last_kernel() == "k_fwd2d_tile", (n0, n1, L, fname, W.last_kernel())
            W.set_option("WL_TILE_LONG", 0)
            y0 = host(W, W.dwt(xd, wt, L))
            W.clear_options()
            assert np.array_equal(y, y0), (n0, n1, L, fname, int((y != y0).sum()))
            W.set_option("WL_TILE_LONG_MAX", 256)                       # the tiles on the two smallest levels only
            y1 = host(W, W.dwt(xd, wt, L))
            W.clear_options()
            assert np.array_equal(y, y1), (n0, n1, L, fname, int((y != y1).sum()))
            assert np.array_equal(y, oracle.dwt_filter(x, wt.qmf, L)), (n0, n1, L, fname)


def test_3d_box_beyond_2_31_elements(gpu, W):
    """A 3-D box whose element offsets do not fit 32 bits (512 x 2048 x 2304 Float32 = 2.4e9 elements, 9.7 GB): the axis / plane
    kernels of the fast 3-D tier against the one-thread-per-output generic tier (different kernels, different index arithmetic:
    device-vs-device, every element), forward and inverse, plus the round trip."""
    import torch
    n = (512, 2048, 2304)
    assert n[0] * n[1] * n[2] > 2 ** 31
    g = torch.Generator(device=gpu).manual_seed(31)
    x = torch.randn(n[2], n[1], n[0], generator=g, dtype=torch.float32, device=gpu).permute(2, 1, 0)      # Julia layout
    assert W.is_julia_layout(x)
    wt = W.wavelet(W.WT.db4)
    y = W.dwt(x, wt, 2)
    assert W.last_kernel() == "k_fwd3d_one", W.last_kernel()                     # (round 6: the one-pass level; 64-bit output offsets)
    try:
        W.set_kernel_path(1)
        yg = W.dwt(x, wt, 2)
        assert W.last_kernel().startswith("k_generic"), W.last_kernel()
    finally:
        W.set_kernel_path(0)
    assert torch.equal(y, yg)
    del yg
    xr = W.idwt(y, wt, 2)
    assert W.last_kernel() == "k_inv3d_one", W.last_kernel()                     # (round 6: the one-pass inverse level)
    try:
        W.set_kernel_path(1)
        xg = W.idwt(y, wt, 2)
    finally:
        W.set_kernel_path(0)
    assert torch.equal(xr, xg)
    del xg, y
    # round trip, in chunks (the Float64 norms of 2.4e9 elements would need 40 GB of temporaries)
    num = den = 0.0
    for k in range(0, n[2], 256):
        a, b = xr[:, :, k:k + 256].double(), x[:, :, k:k + 256].double()
        num += float(((a - b) ** 2).sum())
        den += float((b ** 2).sum())
    assert (num / den) ** 0.5 < 1e-5
    del xr, x
    W.destroy_contexts()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_long_filter_inverse_tiles_bitexact(gpu, W, oracle, dtype):
    """Round 5: the two-level inverse LDS tiles (k_inv2d_tile2, wl_tile.hip) also serve the 12..20-tap filters: the reconstruction
    levels of 128 .. 1024 output rows run two per launch instead of one streaming launch each (and two line passes at 128^2).
    Bit for bit against the oracle and against the per-level tier (transforms_filter.jl:151-155,174-183)."""
    for (n0, n1, L) in ((1024, 1024, 10), (512, 512, 3), (256, 256, 8), (128, 128, 7), (1024, 256, 2), (256, 1024, 8), (2048, 2048, 5)):
        y = rng_array((n0, n1), dtype, n0 + 3 * n1)
        yd = dev(W, y)
        for fname in ("db6", "db7", "sym8", "db9", "db10", "coif4", "beyl"):
            wt = W.wavelet(getattr(W.WT, fname))
            if n0 * n1 > 1 << 20 and fname not in ("sym8", "db10"):
                continue
            x = host(W, W.idwt(yd, wt, L))
            k = W.last_kernel()
            W.set_option("WL_TILE_INV_LONG", 0)
            x0 = host(W, W.idwt(yd, wt, L))
            W.clear_options()
            assert np.array_equal(x, x0), (n0, n1, L, fname, k, int((x != x0).sum()))
            assert np.array_equal(x, oracle.dwt_filter(y, wt.qmf, L, fw=False)), (n0, n1, L, fname, k)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lifting_two_level_tiles_2d(gpu, W, oracle, dtype):
    """Round 5: k_lift2d_tile2_fwd -- two forward 2-D lifting levels per launch on the cache-resident blocks (128 .. 2048 rows): the
    first level's approximation stays in LDS.  Bit for bit against the oracle and against the one-level tiles: the three shipped
    schemes, both element types, odd / even numbers of levels, the 2 x 2 tile grid (cones wrap onto the tile's own block)
    (transforms_lifting.jl:128-194)."""
    for n, Ls in ((128, (1, 2, 3, 7)), (256, (2, 8)), (512, (3, 4)), (1024, (2, 10)), (2048, (5,)), (192, (2, 6)), (4096, (5,))):
        x = rng_array((n, n), dtype, n)
        xd = dev(W, x)
        for sname in ("cdf97", "db2", "haar"):
            if n >= 2048 and sname != "cdf97":
                continue
            sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
            for L in Ls:
                W.set_option("WL_LIFT_TILE2_MAX", 4096)               # (default: Float32 blocks of <= 1024 rows; forced on for every size
                W.set_option("WL_LIFT_TILE2_F64", 1)                  #  and for Float64 here)
                y = host(W, W.dwt(xd, sch, L))
                W.set_option("WL_LIFT_TILE2", 0)
                y0 = host(W, W.dwt(xd, sch, L))
                W.clear_options()
                assert np.array_equal(y, y0), (n, sname, L, int((y != y0).sum()), np.argwhere(y != y0)[:4].tolist())
                if n <= 1024:
                    assert np.array_equal(y, oracle.dwt_lifting(x, sch, L)), (n, sname, L)
                # ... and the two-level inverse tiles (k_lift2d_tile2_inv): the coarser level's result never leaves LDS
                W.set_option("WL_LIFT_TILE2_MAX", 4096)
                W.set_option("WL_LIFT_TILE2_F64", 1)
                xr = host(W, W.idwt(dev(W, y0), sch, L))
                W.set_option("WL_LIFT_TILE2", 0)
                xr0 = host(W, W.idwt(dev(W, y0), sch, L))
                W.clear_options()
                assert np.array_equal(xr, xr0), (n, sname, L, "inverse", int((xr != xr0).sum()), np.argwhere(xr != xr0)[:4].tolist())
                if n <= 1024:
                    assert np.array_equal(xr, oracle.dwt_lifting(y0, sch, L, fw=False)), (n, sname, L, "inverse")
