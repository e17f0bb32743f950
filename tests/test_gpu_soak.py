"""Soak test of the kernel families that place their global loads and `s_waitcnt vmcnt` by hand (wl_dev.h: gload16 / gload4 /
gload16_if / wait_vm / wait_vm_sel): thousands of production-size launches on rotating inputs, every output checksummed on
the device and compared with the checksum of the first transform of the same input.

Why: the hazards of this class (a stale ring register: rounds 2 and 3, DESIGN.md "Hazards") showed up in about one transform
in a thousand, on some boxes only -- the parity tests run each production-size launch once.  The static half of the guard is
tests/test_isa_async_loads.py (CPU); this is the dynamic half.  Any difference between two transforms of the same input is a
failure, whatever the oracle says (the oracle comparison of the same shapes lives in test_gpu_parity.py).
"""
import os

import pytest

pytestmark = pytest.mark.gpu

NLAUNCH = 5000          # per family
NROT = 3                # inputs in rotation (the Infinity Cache holds less than two of the 256 MiB ones)


def _rnd(torch, gen, shape, dtype):
    return torch.randn(shape[1], shape[0], generator=gen, dtype=dtype, device="cuda").t()


def _flat(t):
    # Julia layout = transposed view of a contiguous tensor: the raw buffer
    return t.t() if t.dim() == 2 else t


def _soak_shape(torch, W, shape, dtype, wname, L, expect, **kw):
    """NLAUNCH transforms of NROT inputs in rotation; checksums (wrapping 64-bit sums of the raw bits, taken over the storage
    order of each output) stay on the device until the end."""
    gen = torch.Generator(device="cuda").manual_seed(4242)
    wt = W.wavelet(getattr(W.WT, wname))
    inverse = kw.pop("inverse", False)
    nlaunch = kw.pop("nlaunch", NLAUNCH)
    for k, v in kw.pop("opts", {}).items():
        W.set_option(k, v)
    xs = [_rnd(torch, gen, shape, dtype) for _ in range(NROT)]
    ys = [W.similar(xs[0]) for _ in range(NROT)]
    W.reserve_workspace(xs[0], L, full=True)
    f = W.idwt_ if inverse else W.dwt_
    i64 = torch.int64

    def csum(y):
        return _flat(y).reshape(-1).view(i64).sum()

    ref = torch.zeros(NROT, dtype=i64, device="cuda")
    for r in range(NROT):
        f(ys[r], xs[r], wt, L)
        assert W.last_kernel() == expect, (W.last_kernel(), expect)
        ref[r] = csum(ys[r])
    got = torch.zeros(nlaunch, dtype=i64, device="cuda")
    for it in range(nlaunch):
        r = it % NROT
        f(ys[r], xs[r], wt, L)
        got[it] = csum(ys[r])
    want = ref.repeat((nlaunch + NROT - 1) // NROT)[:nlaunch]
    bad = int((got != want).sum().item())
    assert bad == 0, "%d of %d launches (%s, %s, %s) differ from the first transform of the same input" % (bad, nlaunch, expect, wname, shape)


@pytest.mark.parametrize("wname", ["db4", "sym5", "haar"])
def test_soak_fwd2d_pair(gpu, W, wname):
    """k_fwd2d_pair (8 / 10 / 2 taps): levels 1-2 of the headline shape"""
    import torch
    _soak_shape(torch, W, (8192, 8192), torch.float32, wname, 2, "k_fwd2d_pair")


@pytest.mark.parametrize("wname", ["db4", "sym5"])
def test_soak_fwd2d_lds(gpu, W, wname):
    """k_fwd2d_lds: the single-level member (level 1 of the headline shape)"""
    import torch
    _soak_shape(torch, W, (8192, 8192), torch.float32, wname, 1, "k_fwd2d_lds")


@pytest.mark.parametrize("wname,shape", [("db6", (4096, 4096)), ("db10", (4096, 4096)), ("db8", (2048, 1040))])
def test_soak_fwd2d_lds_long(gpu, W, wname, shape):
    """k_fwd2d_lds_long (12 / 20 / 16 taps; the third shape ends its chunks inside the unrolled body: guarded steps, drain side
    of wait_vm_sel on most steps)"""
    import torch
    _soak_shape(torch, W, shape, torch.float32, wname, 1, "k_fwd2d_lds_long")


@pytest.mark.parametrize("wname,L,expect", [("db4", 2, "k_fwd2d_pair64"), ("haar", 2, "k_fwd2d_pair64"), ("db4", 1, "k_fwd2d_lds64")])
def test_soak_fwd2d_f64(gpu, W, wname, L, expect):
    import torch
    _soak_shape(torch, W, (4096, 4096), torch.float64, wname, L, expect)


def test_soak_inv2d_pair(gpu, W):
    """k_inv2d_pair: levels 2 and 1 of an 8192^2 reconstruction (compiler-placed loads; same marching structure)"""
    import torch
    _soak_shape(torch, W, (8192, 8192), torch.float32, "db4", 2, "k_inv2d_pair", inverse=True, nlaunch=3000)


@pytest.mark.parametrize("wname,dtype,nlaunch", [("sym8", "float32", 2000), ("sym5", "float32", 3000), ("db4", "float64", 1500)])
def test_soak_inv2d_lds_long(gpu, W, wname, dtype, nlaunch):
    """k_inv2d_lds_long (round 4; compiler-placed loads requested 2-3 steps ahead, LDS exchange with alternating buffers and one
    barrier per step): one 8192^2 level of 16 / 10 taps, 4096^2 in Float64"""
    import torch
    dt = getattr(torch, dtype)
    shape = (8192, 8192) if dtype == "float32" else (4096, 4096)
    _soak_shape(torch, W, shape, dt, wname, 1, "k_inv2d_lds_long", inverse=True, nlaunch=nlaunch)


def test_soak_ti_denoise_batch(gpu, W):
    """the plane-batched (BT) instances of the pair / level kernels: a translation-invariant denoise of 2048^2, 4 x 4 spins"""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(99)
    wt = W.wavelet(W.WT.sym5)
    xs = [_rnd(torch, gen, (2048, 2048), torch.float32) for _ in range(NROT)]
    i64 = torch.int64
    ref = []
    for x in xs:
        y = W.denoise(x, wt, L=4, TI=True, nspin=(4, 4))
        ref.append(_flat(y).reshape(-1).view(i64).sum())
    n = 400
    got = torch.zeros(n, dtype=i64, device="cuda")
    for it in range(n):
        y = W.denoise(xs[it % NROT], wt, L=4, TI=True, nspin=(4, 4))
        got[it] = _flat(y).reshape(-1).view(i64).sum()
    want = torch.stack(ref).repeat((n + NROT - 1) // NROT)[:n]
    assert int((got != want).sum().item()) == 0


@pytest.mark.parametrize("wname,inverse,expect", [("db4", False, "k_fwd3d_one"), ("db2", True, "k_inv3d_one"), ("db3", False, "k_fwd3d_one")])
def test_soak_3d_one_pass_levels(gpu, W, wname, inverse, expect):
    """Round 6: the one-pass 3-D levels (k_fwd3d_one: landing ring of F + 2 planes tied to the dim-3 sums, vmcnt(F + 1);
    k_inv3d_one: two rounds of SH + 2 columns in flight, vmcnt(SH + 2)) on a 512 x 512 x 128 box (256 MiB in rotation x 3)."""
    import torch
    gen = torch.Generator(device="cuda").manual_seed(777)
    wt = W.wavelet(getattr(W.WT, wname))
    xs = [torch.randn(128, 512, 512, generator=gen, dtype=torch.float32, device="cuda").permute(2, 1, 0) for _ in range(NROT)]
    ys = [W.similar(xs[0]) for _ in range(NROT)]
    W.reserve_workspace(xs[0], 1, full=True)
    f = W.idwt_oop_ if inverse else W.dwt_oop_
    i64 = torch.int64

    def csum(y):
        return y.permute(2, 1, 0).reshape(-1).view(i64).sum()          # (the raw buffer: Julia layout = permuted view of a contiguous tensor)

    nlaunch = int(os.environ.get("WL_SOAK_3D_LAUNCHES", "1500"))
    ref = torch.zeros(NROT, dtype=i64, device="cuda")
    for r in range(NROT):
        f(ys[r], xs[r], wt, 1)
        assert W.last_kernel() == expect, (W.last_kernel(), expect)
        ref[r] = csum(ys[r])
    got = torch.zeros(nlaunch, dtype=i64, device="cuda")
    for it in range(nlaunch):
        r = it % NROT
        f(ys[r], xs[r], wt, 1)
        got[it] = csum(ys[r])
    want = ref.repeat((nlaunch + NROT - 1) // NROT)[:nlaunch]
    bad = int((got != want).sum().item())
    assert bad == 0, "%d of %d launches (%s, %s) differ from the first transform of the same input" % (bad, nlaunch, expect, wname)
