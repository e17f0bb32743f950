"""Differential tests of the dispatch tiers against each other, on the device: every kernel family added for speed must give the
bits of the family it replaces, on many more random shapes / depths / chunk lengths than the oracle-based parity tests can afford
(those pin each family to the CPU oracle on a handful of shapes; a device-vs-device comparison costs milliseconds per case).
Test infrastructure, not product code."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rnd(torch, gen, shape, dtype):
    # Julia (column-major) layout: a transposed view of a C-contiguous tensor
    return torch.randn(shape[1], shape[0], generator=gen, dtype=dtype).cuda().t()


@pytest.mark.parametrize("seed", [1, 2])
def test_inverse_pair_against_single_level_launches(gpu, W, seed):
    import torch
    r = np.random.default_rng(seed)
    gen = torch.Generator(device="cpu").manual_seed(100 + seed)
    hit = 0
    for _ in range(60):
        n0 = int(r.choice([1024, 1536, 2048, 2560, 3072, 4096]))
        n1 = int(r.choice([64, 96, 160, 256, 512, 1056, 2048]))
        if n0 * n1 > (1 << 23):
            n1 = 256
        wt = W.wavelet(getattr(W.WT, str(r.choice(["haar", "db2", "db3", "db4", "sym4"]))))
        x = _rnd(torch, gen, (n0, n1), torch.float32)
        L = int(r.integers(2, W.maxtransformlevels(x) + 1))
        out = []
        for on in (0, 1):
            W.set_option("WL_INV_PAIR", on)
            W.set_option("WL_INV_PAIR_MIN", 0)
            W.set_option("WL_TILE_INV", int(r.integers(0, 2)) if on else 0)
            W.set_option("WL_INVPAIR_TP", int(r.choice([16, 32, 64])))
            out.append(W.idwt(x, wt, L))
            hit += int(on and W.last_kernel() == "k_inv2d_pair")
        W.clear_options()
        assert torch.equal(out[0], out[1]), (n0, n1, L)
    assert hit >= 20            # (the comparison must not be vacuous)


@pytest.mark.parametrize("seed", [3, 4])
def test_lifting_tiles_against_marching_kernels(gpu, W, seed):
    import torch
    r = np.random.default_rng(seed)
    gen = torch.Generator(device="cpu").manual_seed(100 + seed)
    hit = 0
    for _ in range(60):
        n = int(r.choice([128, 192, 256, 320, 384, 512, 640, 1024, 1280, 2048]))
        dt = torch.float32 if r.random() < 0.7 else torch.float64
        sch = W.wavelet(getattr(W.WT, str(r.choice(["cdf97", "db2", "haar"]))), W.WT.Lifting)
        x = _rnd(torch, gen, (n, n), dt)
        L = int(r.integers(1, W.maxtransformlevels(x) + 1))
        out = []
        for on in (0, 1):
            W.set_option("WL_LIFT_TILE", on)
            W.set_option("WL_LIFT_TP", int(r.choice([0, 8, 24, 40, 64])))
            y = W.dwt(x, sch, L)
            hit += int(on and W.last_kernel() == "k_lift2d_tile")
            out.append((y, W.idwt(y, sch, L)))
        W.clear_options()
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), (n, dt, L)
    assert hit >= 20


@pytest.mark.parametrize("seed", [5])
def test_forward_tile_and_pair_tiers_against_single_levels(gpu, W, seed):
    import torch
    r = np.random.default_rng(seed)
    gen = torch.Generator(device="cpu").manual_seed(100 + seed)
    seen = set()
    for _ in range(60):
        n0 = int(r.choice([512, 1024, 2048, 4096]))
        n1 = int(r.choice([256, 512, 1024, 2048]))
        wt = W.wavelet(getattr(W.WT, str(r.choice(["haar", "db2", "db3", "db4", "sym5"]))))
        x = _rnd(torch, gen, (n0, n1), torch.float32)
        L = int(r.integers(2, W.maxtransformlevels(x) + 1))
        out = []
        pair = r.random() < 0.5          # this case: the fused pair (which the no-staging tile kernel would pre-empt) or that tile kernel
        for on in (0, 1):
            W.set_option("WL_TILEB", int(on and not pair))
            W.set_option("WL_TILEB_MIN", 0)
            W.set_option("WL_TILEB_MAX", 4096)
            W.set_option("WL_LDS_PAIR_MIN", 0 if (on and pair) else (1 << 62))
            out.append(W.dwt(x, wt, L))
            if on:
                seen.add(W.last_kernel())
        W.clear_options()
        assert torch.equal(out[0], out[1]), (n0, n1, L)
    assert "k_fwd2d_tileB" in seen and "k_fwd2d_pair" in seen


def test_ti_denoise_fused_tiers_against_separate_passes(gpu, W):
    """batched fused pair + thresholds applied at the stores, against level-by-level launches and the separate threshold pass"""
    import torch
    r = np.random.default_rng(6)
    gen = torch.Generator(device="cpu").manual_seed(106)
    for _ in range(16):
        n = int(r.choice([512, 1024]))
        wt = W.wavelet(getattr(W.WT, str(r.choice(["db2", "db4", "sym5"]))))
        x = _rnd(torch, gen, (n, n), torch.float32)
        nspin = (int(r.integers(1, 5)), int(r.integers(1, 5)))
        L = int(r.integers(1, 7))
        out = []
        for on in (0, 1):
            W.set_option("WL_PAIR_BATCH", on)
            W.set_option("WL_PAIR_BATCH_MIN", 0)
            W.set_option("WL_TI_FUSE_TH", on)
            out.append(W.denoise(x, wt, L=L, TI=True, nspin=nspin))
        W.clear_options()
        assert torch.equal(out[0], out[1]), (n, nspin, L)


@pytest.mark.parametrize("seed", [7, 8])
def test_wpt_fast_tiers_against_per_depth_kernels(gpu, W, seed):
    """round 4: k_wpt_fwd_multi / k_wpt_fwd_tail / k_wpt_inv_tail and the ping-ponged lifting depths against the per-depth tier
    (WL_WPT_FAST = 0) on random lengths (powers of two and 3 * 2^k), depths, filters, element types and random valid trees"""
    import torch
    r = np.random.default_rng(seed)
    gen = torch.Generator(device="cpu").manual_seed(200 + seed)
    hits = set()
    for it in range(70):
        lg = int(r.integers(6, 19))
        n = (1 << lg) if r.random() < 0.8 else 3 * (1 << (lg - 2))
        dt = torch.float32 if r.random() < 0.7 else torch.float64
        x = torch.randn(n, generator=gen, dtype=dt).cuda()
        Lmax = W.maxtransformlevels(n)
        kind = r.random()
        if kind < 0.55:
            tree = int(r.integers(1, Lmax + 1))                       # full tree by depth (wl_wpt_*_full)
        elif kind < 0.8:
            tree = W.maketree(n, int(r.integers(1, Lmax + 1)), "full")    # the same through a tree vector
        else:
            t = np.zeros(2 ** Lmax - 1, dtype=np.uint8)
            t[0] = 1
            p = float(r.choice([0.5, 0.8, 0.95]))
            for i in range(1, len(t)):
                t[i] = 1 if (t[(i + 1) // 2 - 1] and r.random() < p) else 0
            tree = t
        lifting = r.random() < 0.3
        wt = (W.wavelet(getattr(W.WT, str(r.choice(["cdf97", "db2", "haar"]))), W.WT.Lifting) if lifting
              else W.wavelet(getattr(W.WT, str(r.choice(["haar", "db2", "db3", "db4", "sym5"])))))
        out = []
        for fast in (0, 1):
            W.set_option("WL_WPT_FAST", fast)
            y = W.wpt(x, wt, tree)
            if fast:
                hits.add(W.last_kernel())
            xr = W.iwpt(y, wt, tree)
            if fast:
                hits.add(W.last_kernel())
            out.append((y, xr))
        W.clear_options()
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), (n, dt, lifting, type(tree), it)
    assert {"k_wpt_fwd_multi", "k_wpt_fwd_tail", "k_wpt_inv_tail", "k_wpt_inv_multi"} <= hits, hits


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_inverse_lds_exchange_level_against_the_tiers_it_replaces(gpu, W, seed):
    """k_inv2d_lds_long (wl_inv2d_long.hip) against the two-pass long-filter tier / the DPP streaming kernel: random row counts
    (multiples of 256), column counts (any even number that fits the reach), depths, strip widths, chunk lengths, request
    distances, both element types, single images and batches of planes."""
    import torch
    r = np.random.default_rng(seed)
    gen = torch.Generator(device="cpu").manual_seed(300 + seed)
    hit = 0
    names = ["db4", "sym5", "db6", "sym7", "db8", "coif4", "db9", "coif6", "db10", "beyl"]
    for it in range(48):
        n0 = 256 * int(r.integers(1, 9))
        n1 = 2 * int(r.integers(16, 600))
        dt = torch.float32 if r.random() < 0.65 else torch.float64
        wt = W.wavelet(getattr(W.WT, str(r.choice(names))))
        nb = int(r.choice([0, 0, 2, 5]))                      # 0: a single image
        if n0 * n1 * max(nb, 1) > (1 << 22):
            n1 = 2 * int(r.integers(16, 64))
        if nb:
            x = torch.randn(nb, n1, n0, generator=gen, dtype=dt).cuda().permute(2, 1, 0)
        else:
            x = _rnd(torch, gen, (n0, n1), dt)
        Lmax = min(W.maxtransformlevels(x[:, :, 0] if nb else x), 4)
        if Lmax < 1:
            continue
        L = int(r.integers(1, Lmax + 1))
        out = []
        for on in (0, 1):
            W.clear_options()
            W.set_option("WL_INVLONG2D", on)
            W.set_option("WL_INVLONG2D_MIN_ROWS", 256)
            W.set_option("WL_INVLONG_SHORT_MIN", 0)
            W.set_option("WL_INVLONG_FMIN", 8)
            W.set_option("WL_INV_PAIR", 0)
            W.set_option("WL_TILE_INV", 0)
            W.set_option("WL_INVLONG_W", int(r.choice([0, 1, 2, 4])))
            W.set_option("WL_INVLONG_TP", int(r.choice([8, 12, 16, 24, 40, 64, 128])))
            W.set_option("WL_INVLONG_D", int(r.choice([1, 2, 3, 4])))
            W.set_option("WL_INVLONG_PPL", int(r.choice([0, 1, 2])))
            W.set_option("WL_INVLONG_WAVES_PER_CU", int(r.choice([0, 8])))
            out.append(W.idwt_batch(x, wt, L) if nb else W.idwt(x, wt, L))
            hit += int(on and W.last_kernel() == "k_inv2d_lds_long")
        W.clear_options()
        assert torch.equal(out[0], out[1]), (n0, n1, nb, L, str(dt), wt.name)
    assert hit >= 15            # (the comparison must not be vacuous)


@pytest.mark.parametrize("seed", [10, 11])
def test_lifting_3d_lds_tail_against_per_axis_launches(gpu, W, seed):
    import torch
    r = np.random.default_rng(seed)
    gen = torch.Generator(device="cpu").manual_seed(400 + seed)
    for it in range(40):
        n = int(r.choice([8, 16, 32, 64, 128]))
        dt = torch.float32 if r.random() < 0.6 else torch.float64
        sch = W.wavelet(getattr(W.WT, str(r.choice(["cdf97", "db2", "haar"]))), W.WT.Lifting)
        x = torch.randn(n, n, n, generator=gen, dtype=dt).cuda().permute(2, 1, 0)
        L = int(r.integers(1, W.maxtransformlevels(x) + 1))
        ys, xs = [], []
        for on in (0, 1):
            W.set_option("WL_LIFT_TAIL3D", on)
            y = W.dwt(x, sch, L)
            ys.append(y)
            xs.append(W.idwt(y, sch, L))
        W.clear_options()
        assert torch.equal(ys[0], ys[1]) and torch.equal(xs[0], xs[1]), (n, L, str(dt), sch.name)


@pytest.mark.parametrize("seed", [21, 22])
def test_3d_one_pass_and_block_levels_against_axis_passes(gpu, W, seed):
    """Round 6: k_fwd3d_one (one forward 3-D level per pass over HBM) and k_level3_lds (small levels in one launch, both directions)
    against the tiers they replace (WL_3D_ONE = 0, WL_LEVEL3 = 0: single-axis passes / plane kernels / any-extent kernels), random
    boxes, both element types, 2 ... 10 taps, random depths and segment lengths; device against device, every element."""
    import torch
    r = np.random.default_rng(seed)
    gen = torch.Generator(device="cpu").manual_seed(300 + seed)
    hit1 = hit3 = hiti = 0
    for it in range(36):
        dt = torch.float32 if r.random() < 0.6 else torch.float64
        if it % 2 == 0:          # streaming sizes: lines of 128 ... 1024
            n0 = int(r.choice([128, 256, 256, 512, 1024])) if r.random() < 0.5 else 4 * int(r.integers(8, 257))
            n1 = 2 * int(r.integers(8, 100))
            n2 = 2 * int(r.integers(8, 50))
            while n0 * n1 * n2 > (1 << 23):
                n1 = max(16, (n1 // 4) * 2)
        else:                    # block sizes: any even extents
            n0, n1, n2 = (2 * int(r.integers(8, 49)) for _ in range(3))
        fname = str(r.choice(["haar", "db2", "db3", "db4", "db4", "sym4", "db5", "sym5"]))
        wt = W.wavelet(getattr(W.WT, fname))
        x = torch.randn(n2, n1, n0, generator=gen, dtype=dt).cuda().permute(2, 1, 0)        # Julia layout
        L = int(r.integers(1, min(3, W.maxtransformlevels(x)) + 1))
        W.set_option("WL_3D_ONE", 0)
        W.set_option("WL_I3D_ONE", 0)
        W.set_option("WL_LEVEL3", 0)
        y0 = W.dwt(x, wt, L)
        x0 = W.idwt(y0, wt, L)
        W.clear_options()
        W.set_option("WL_3D_ONE_MIN", 0)
        W.set_option("WL_3D_ONE_TJ", int(r.choice([8, 16, 32, 64])))
        W.set_option("WL_3D_ONE_WAVES", int(r.choice([0, 8])))
        for key in ("WL_I3D_ONE_MIN", "WL_I3D_ONE_MIN_LONG", "WL_I3D_ONE_MIN_ANY"):
            W.set_option(key, 0)
        W.set_option("WL_I3D_ONE_F64_FMAX", 8)
        W.set_option("WL_I3D_ONE_TK", int(r.choice([4, 8, 16, 32])))
        W.set_option("WL_I3D_ONE_WAVES", int(r.choice([0, 8])))
        y1 = W.dwt(x, wt, L)
        kf = W.last_kernel()
        x1 = W.idwt(y0, wt, L)
        ki = W.last_kernel()
        W.clear_options()
        hit1 += int(kf == "k_fwd3d_one")
        hiti += int(ki == "k_inv3d_one")
        hit3 += int(kf == "k_level3_lds") + int(ki == "k_level3_lds")
        assert torch.equal(y0, y1), (n0, n1, n2, fname, L, kf, str(dt))
        assert torch.equal(x0, x1), (n0, n1, n2, fname, L, ki, str(dt))
    assert hit1 >= 8 and hit3 >= 4 and hiti >= 6, (hit1, hit3, hiti)            # (the comparison must not be vacuous)
