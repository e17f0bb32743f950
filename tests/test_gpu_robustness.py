"""Robustness of the vectorised GPU backend (checks the reference does not need, a kernel family of 16-byte loads / stores does):

* guard bands: every array handed to the library sits between 4 KiB canaries inside one device buffer; after the call the
  canaries and the input must be untouched -- a bit-exact comparison of the in-bounds region cannot see a 16-byte store that
  lands one vector past the end;
* hipGraph: a transform captured into a graph (torch.cuda.CUDAGraph == hipGraph on ROCm) and replayed on new input data must
  give the bits of a direct call -- INTEGRATION.md promises that calls are capturable once the workspace is reserved;
* two devices: a call on device 1 leaves the caller's current device alone (runs where a second GPU is visible).
"""
import numpy as np
import pytest

from conftest import rng_array

pytestmark = pytest.mark.gpu

CANARY = 1024            # elements on either side of every array (4 KiB of f32, 8 KiB of f64)


class Guarded:
    """arrays carved out of ONE device buffer, each between canaries of a NaN bit pattern"""

    def __init__(self, torch, dtype, sizes, misalign=0):
        self.torch = torch
        self.td = torch.float32 if dtype == np.float32 else torch.float64
        total = CANARY + sum(int(n) + CANARY for n in sizes) + 64
        self.buf = torch.empty(total, dtype=self.td, device="cuda")
        self.pattern = float("nan")
        self.buf.fill_(self.pattern)
        self.offs, off = [], CANARY + int(misalign)
        for n in sizes:
            self.offs.append((off, int(n)))
            off += int(n) + CANARY
        self.snapshot = None

    def flat(self, i):
        off, n = self.offs[i]
        return self.buf[off:off + n]

    def jl(self, i, shape):
        """Julia-layout (column-major) view of array i"""
        t = self.flat(i).view(*reversed(shape))
        return t.permute(*reversed(range(len(shape)))) if len(shape) > 1 else t

    def arm(self):
        self.snapshot = self.buf.clone()

    def check(self, written, label):
        """everything outside the arrays listed in `written` must be bit-identical to the snapshot"""
        torch = self.torch
        same = (self.buf.view(torch.int32 if self.td == torch.float32 else torch.int64) ==
                self.snapshot.view(torch.int32 if self.td == torch.float32 else torch.int64))
        keep = torch.ones_like(same)
        for i in written:
            off, n = self.offs[i]
            keep[off:off + n] = False
        bad = (~same) & keep
        if bool(bad.any()):
            idx = torch.nonzero(bad).flatten()[:8].tolist()
            where = [(j, [k for k, (o, n) in enumerate(self.offs) if o - CANARY <= j < o + n + CANARY]) for j in idx]
            raise AssertionError(f"{label}: {int(bad.sum())} elements outside the output were modified, first at (offset, near array) {where}")


def _put(g, i, shape, a, W):
    g.jl(i, shape).copy_(W.to_device(a))


FILTER_CASES = [
    # (label, shape, filter, L, dtype, options)  -- one per forward / inverse kernel family of the filter-bank path
    ("2-D pair + lds + tile + tail", (4096, 4096), "db4", 12, np.float32, {}),
    ("2-D lds single levels", (2048, 1024), "sym5", 3, np.float32, {"WL_FUSE2": 0, "WL_TILE": 0, "WL_M2D_MAX": 64}),
    ("2-D tile + tail", (1024, 1024), "db2", 10, np.float32, {}),
    ("2-D multi", (256, 512), "db3", 3, np.float32, {"WL_TILE": 0}),
    ("2-D any even size (gtile)", (1080, 1920), "db4", 3, np.float32, {}),
    ("2-D odd-sized level (any-axis)", (1000, 600), "db4", 3, np.float32, {}),
    ("2-D f64 stream + tile", (2048, 2048), "db4", 11, np.float64, {}),
    ("2-D 16 taps (long lines + axis)", (2048, 2048), "sym8", 5, np.float32, {}),
    ("2-D battle 23 taps (vlong)", (1024, 1024), "batt2", 4, np.float32, {}),
    ("2-D generic path", (192, 320), "db4", 3, np.float32, {"path": 1}),
    ("1-D multi + tail", (1 << 20,), "db4", 20, np.float32, {}),
    ("1-D f64", (1 << 18,), "db2", 18, np.float64, {}),
    ("1-D any even length", (1000000,), "db4", 6, np.float32, {}),
    ("1-D long filter", (1 << 18,), "db8", 10, np.float32, {}),
    ("3-D axis kernels + tail3", (128, 128, 128), "db4", 7, np.float32, {}),
    ("3-D odd box", (100, 60, 36), "db2", 2, np.float32, {}),
    ("3-D one-pass levels, Float64, partly filled waves", (200, 48, 40), "db3", 2, np.float64, {"WL_3D_ONE_MIN_ANY": 0, "WL_I3D_ONE_MIN_ANY": 0, "WL_I3D_ONE_F64_FMAX": 8}),
    ("3-D one-pass levels, 8-byte lanes", (300, 26, 30), "db4", 1, np.float32, {"WL_3D_ONE_MIN_ANY": 0, "WL_I3D_ONE_MIN_ANY": 0}),
]


@pytest.mark.parametrize("misalign", [0, 1])
def test_guard_bands_filter(gpu, W, oracle, misalign):
    """forward and inverse filter-bank transforms of every kernel family: result bit-exact, canaries and input untouched.
    misalign = 1 shifts every array by one element off the 16-byte grid (the vector kernels must decline or cope)."""
    import torch
    for label, shape, fname, L, dtype, opts in FILTER_CASES:
        W.clear_options()
        W.set_kernel_path(opts.get("path", 0))
        for k, v in opts.items():
            if k != "path":
                W.set_option(k, v)
        wt = W.wavelet(getattr(W.WT, fname))
        N = int(np.prod(shape))
        a = rng_array(shape, dtype, N % 9973 + misalign)
        g = Guarded(torch, dtype, [N, N, N], misalign)
        _put(g, 0, shape, a, W)
        x, y, z = g.jl(0, shape), g.jl(1, shape), g.jl(2, shape)
        g.arm()
        W.dwt_oop_(y, x, wt, L)
        torch.cuda.synchronize()
        g.check([1], f"dwt {label} misalign={misalign}")
        ye = oracle.dwt_filter(a, wt.qmf, L)
        assert np.array_equal(W.to_host(y), ye), (label, misalign, W.last_kernel())
        g.arm()
        W.idwt_oop_(z, y, wt, L)
        torch.cuda.synchronize()
        g.check([2], f"idwt {label} misalign={misalign}")
        assert np.array_equal(W.to_host(z), oracle.dwt_filter(ye, wt.qmf, L, fw=False)), (label, misalign, W.last_kernel())
    W.set_kernel_path(0)


LIFT_CASES = [
    ("1-D lifting fused + register tail", (1 << 20,), "cdf97", 20, np.float32),
    ("1-D lifting f64", (1 << 16,), "cdf97", 16, np.float64),
    ("1-D lifting any even length", (300000,), "cdf97", 5, np.float32),
    ("2-D lifting fused levels + tails", (2048, 2048), "cdf97", 11, np.float32),
    ("2-D lifting any even size", (1000, 1000), "cdf97", 3, np.float32),
    ("2-D lifting db2", (512, 512), "db2", 9, np.float32),
    ("3-D lifting cube", (64, 64, 64), "cdf97", 6, np.float32),
]


@pytest.mark.parametrize("misalign", [0, 1])
def test_guard_bands_lifting(gpu, W, oracle, misalign):
    """lifting transforms: in place (dwt!(y, scheme)) and out of place (dwt_oop!), forward and inverse"""
    import torch
    for label, shape, sname, L, dtype in LIFT_CASES:
        sch = W.wavelet(getattr(W.WT, sname), W.WT.Lifting)
        N = int(np.prod(shape))
        a = rng_array(shape, dtype, N % 9973 + 5 + misalign)
        g = Guarded(torch, dtype, [N, N], misalign)
        _put(g, 0, shape, a, W)
        x, y = g.jl(0, shape), g.jl(1, shape)
        g.arm()
        W.dwt_oop_(y, x, sch, L)
        torch.cuda.synchronize()
        g.check([1], f"lifting dwt_oop {label} misalign={misalign}")
        ye = oracle.dwt_lifting(a, sch, L)
        assert np.array_equal(W.to_host(y), ye), (label, misalign, W.last_kernel())
        g.arm()
        W.idwt_(y, sch, L)                       # in place
        torch.cuda.synchronize()
        g.check([1], f"lifting idwt! {label} misalign={misalign}")
        assert np.array_equal(W.to_host(y), oracle.dwt_lifting(ye, sch, L, fw=False)), (label, misalign, W.last_kernel())
        g.arm()
        W.dwt_(x, sch, L)                        # in place, forward
        torch.cuda.synchronize()
        g.check([0], f"lifting dwt! {label} misalign={misalign}")
        assert np.array_equal(W.to_host(x), ye), (label, misalign, W.last_kernel())


def test_guard_bands_batched_and_callers(gpu, W, oracle):
    """dwtc with a leading dimension larger than the signal length (odd pitch: unaligned columns), wpt, modwt, threshold"""
    import torch
    db4 = W.wavelet(W.WT.db4)
    for n, ns, ld in ((4096, 48, 4096), (4096, 33, 4099), (1 << 16, 16, (1 << 16) + 8), (1000, 7, 1000)):
        a = rng_array((n, ns), np.float32, n + ns + ld)
        g = Guarded(torch, np.float32, [ld * ns, ld * ns])
        xs = torch.as_strided(g.flat(0), (n, ns), (1, ld))
        ys = torch.as_strided(g.flat(1), (n, ns), (1, ld))
        xs.copy_(W.to_device(a))
        g.arm()
        L = W.maxtransformlevels(n)
        # the raw ABI: the host mirror only passes dense matrices (ld = len)
        import ctypes as C
        from wavelets_jl_amd import transforms as TR
        h, st = TR._context(g.buf.device)
        q = np.ascontiguousarray(db4.qmf, dtype=np.float64)
        rc = W._lib.load().wl_dwtc_filter(h, 0, C.c_void_p(ys.data_ptr()), C.c_void_p(xs.data_ptr()), n, ns, ld,
                                          q.ctypes.data_as(C.POINTER(C.c_double)), len(q), L, 1, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        # only the n x ns window of the pitched output may change: rows n .. ld-1 of every column are guard space as well
        keep = g.buf.clone()
        got = W.to_host(ys)
        assert np.array_equal(got, oracle.dwtc_filter(a, db4.qmf, L)), (n, ns, ld, W.last_kernel())
        ys.copy_(torch.as_strided(g.snapshot, (n, ns), (1, ld), g.offs[1][0]))     # put the armed content back into the window ...
        g.check([], f"dwtc n={n} ns={ns} ld={ld}")                                  # ... then nothing at all may differ
        g.buf.copy_(keep)
    # packet transform (full tree), modwt, threshold on guarded vectors
    n = 1 << 14
    a = rng_array((n,), np.float32, 77)
    g = Guarded(torch, np.float32, [n, n, n * 5])
    _put(g, 0, (n,), a, W)
    x, y = g.jl(0, (n,)), g.jl(1, (n,))
    g.arm()
    W.wpt_(y, x, db4, 4)
    torch.cuda.synchronize()
    g.check([1], "wpt")
    assert np.array_equal(W.to_host(y), oracle.wpt_filter(a, db4.qmf, W.maketree(n, 4, "full")))
    g.arm()
    W.threshold_(y, W.HardTH(), 0.5)
    torch.cuda.synchronize()
    g.check([1], "threshold")


def test_nonsense_option_values_are_harmless(gpu, W, oracle):
    """The per-context options are tuning / test knobs, but a zero or an odd value must not be able to crash the process (chunk and
    tile lengths are divisors in the launchers): every knob that sizes a chunk, a tile or a slab set to 0, 1 and 7 -- the
    transforms still run and still give the oracle's bits."""
    knobs = ("WL_TJ", "WL_TJ2", "WL_TS", "WL_INV2D_TP", "WL_SLAB_LINES", "WL_LONG_TJ", "WL_LIFT_TP", "WL_TAIL2_THREADS", "WL_PAIR_W",
             "WL_PAIR_W64", "WL_LONG_W", "WL_TILEB_MAX", "WL_LIFT_TILE_MAX")
    db4, db8, cdf = W.wavelet(W.WT.db4), W.wavelet(W.WT.db8), W.wavelet(W.WT.cdf97, W.WT.Lifting)
    cases = [((512, 512), np.float32, db4, 3, False), ((512, 256), np.float64, db4, 2, False), ((1 << 16,), np.float32, db4, 16, False),
             ((512, 512), np.float32, db8, 2, False), ((512, 512), np.float32, cdf, 3, True), ((1 << 15,), np.float64, cdf, 15, True)]
    data = []
    for shape, dtype, wt, L, lifting in cases:
        x = rng_array(shape, dtype, 3 + len(shape))
        ye = oracle.dwt_lifting(x, wt, L) if lifting else oracle.dwt_filter(x, wt.qmf, L)
        data.append((x, ye))
    xc = rng_array((4096, 6), np.float32, 9)
    yc = oracle.dwtc_filter(xc, db4.qmf, 5)
    for bad in (0, 1, 7):
        for k in knobs:
            W.set_option(k, bad)
        W.set_option("WL_LDS_PAIR_MIN", 0)
        try:
            for (shape, dtype, wt, L, lifting), (x, ye) in zip(cases, data):
                y = W.to_host(W.dwt(W.to_device(x), wt, L))
                assert np.array_equal(y, ye), (bad, shape, dtype, L)
                xr = W.to_host(W.idwt(W.to_device(ye), wt, L))
                xe = oracle.dwt_lifting(ye, wt, L, fw=False) if lifting else oracle.dwt_filter(ye, wt.qmf, L, fw=False)
                assert np.array_equal(xr, xe), (bad, shape, dtype, L, "inv")
            assert np.array_equal(W.to_host(W.dwtc(W.to_device(xc), db4, 5)), yc), bad
        finally:
            W.clear_options()


def test_hipgraph_capture_and_replay(gpu, W, oracle):
    """A transform call allocates nothing and synchronises nothing once its context exists and the workspace is reserved, so it can
    be captured into a hipGraph; replays on new input data give the bits of a direct call."""
    import torch
    cases = [((1024, 1024), np.float32, W.wavelet(W.WT.db4), 10, False),
             ((4096, 4096), np.float32, W.wavelet(W.WT.db4), 12, False),
             ((1 << 18,), np.float64, W.wavelet(W.WT.db2), 18, False),
             ((1 << 18,), np.float32, W.wavelet(W.WT.cdf97, W.WT.Lifting), 18, True),
             ((512, 512), np.float32, W.wavelet(W.WT.cdf97, W.WT.Lifting), 9, True),
             ((256, 128, 64), np.float32, W.wavelet(W.WT.db4), 5, False),           # round 6: k_fwd3d_one + k_level3_lds + k_tail3
             ((128, 64, 256), np.float64, W.wavelet(W.WT.db2), 4, False)]
    s = torch.cuda.Stream()
    for shape, dtype, wt, L, lifting in cases:
        inputs = [rng_array(shape, dtype, 11 + k) for k in range(3)]
        x = W.to_device(inputs[0])
        y = W.similar(x)
        with torch.cuda.stream(s):
            W.reserve_workspace(x, L, full=True)         # the context of stream s and all the scratch any path may want
            W.dwt_oop_(y, x, wt, L)                      # (first call: code objects loaded, nothing left to allocate)
        torch.cuda.synchronize()
        held = W.workspace_held() if hasattr(W, "workspace_held") else None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            W.dwt_oop_(y, x, wt, L)
        for k in (1, 2, 0):
            x.copy_(W.to_device(inputs[k]))
            y.zero_()
            graph.replay()
            torch.cuda.synchronize()
            e = oracle.dwt_lifting(inputs[k], wt, L) if lifting else oracle.dwt_filter(inputs[k], wt.qmf, L)
            assert np.array_equal(W.to_host(y), e), (shape, dtype, L, k)
        del graph
    # wavelet packets (round 4: no stream synchronisation inside wl_wpt_*): full trees are capturable
    for n, dtype, wt, L in (((1 << 18), np.float32, W.wavelet(W.WT.db4), 6), ((1 << 14), np.float32, W.wavelet(W.WT.db4), 14),
                            ((1 << 16), np.float32, W.wavelet(W.WT.cdf97, W.WT.Lifting), 4)):
        lifting = not hasattr(wt, "qmf")
        inputs = [rng_array((n,), dtype, 31 + k) for k in range(3)]
        tree = W.maketree(n, L, "full")
        x = W.to_device(inputs[0])
        with torch.cuda.stream(s):
            W.reserve_workspace(x, L, full=True)
            y = W.wpt(x, wt, tree)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            if lifting:
                y.copy_(x)
                W.wpt_(y, wt, tree)
            else:
                W.wpt_(y, x, wt, tree)
        for k in (1, 2, 0):
            x.copy_(W.to_device(inputs[k]))
            graph.replay()
            torch.cuda.synchronize()
            e = oracle.wpt_lifting(inputs[k], wt, tree) if lifting else oracle.wpt_filter(inputs[k], wt.qmf, tree)
            assert np.array_equal(W.to_host(y), e), (n, L, k)
        del graph


def test_calls_leave_the_current_device_alone_two_gpus(gpu, W, oracle):
    """runs where two GPUs are visible: a transform of a tensor on device 1 while device 0 is current"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    a = rng_array((512, 512), np.float32, 3)
    wt = W.wavelet(W.WT.db4)
    torch.cuda.set_device(0)
    x1 = W.to_device(a, device="cuda:1")
    y1 = W.dwt(x1, wt, 4)
    torch.cuda.synchronize(1)
    assert torch.cuda.current_device() == 0
    assert np.array_equal(W.to_host(y1), oracle.dwt_filter(a, wt.qmf, 4))
