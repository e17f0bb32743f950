"""Host-side mirror of the reference interface (WT types, Util helpers, argument contract).
No GPU needed: these tests stop at the ABI boundary."""
import os
import re

import numpy as np
import pytest


def test_wavelet_constructors(W):
    WT, wavelet = W.WT, W.wavelet
    f = wavelet(WT.db4)
    assert isinstance(f, W.OrthoFilter) and len(f) == 8 and f.name == "db4"
    assert isinstance(wavelet(WT.db4, WT.Filter), W.OrthoFilter)
    assert isinstance(wavelet(WT.db4, WT.Filter, WT.Periodic), W.OrthoFilter)
    g = wavelet(WT.cdf97, WT.Lifting)
    assert isinstance(g, W.GLS) and g.name == "cdf9/7" and len(g.step) == 4
    assert [type(s.steptype).__name__ for s in g.step] == ["UpdateStep", "PredictStep", "UpdateStep", "PredictStep"]
    assert g.norm1 == 1.1496043988603355 and g.norm2 == 0.8698644516247099
    # cdf9/7 exists only as a lifting scheme (wt_main.jl:262-264)
    with pytest.raises(TypeError):
        wavelet(WT.cdf97)
    with pytest.raises(TypeError):
        wavelet(WT.cdf97, WT.Filter)
    with pytest.raises(ValueError):
        wavelet(WT.db4, WT.Lifting)          # "scheme not found"
    with pytest.raises(ValueError):
        wavelet(WT.Coiflet(12))              # "filter not found"
    with pytest.raises(TypeError):
        wavelet("db4")


def test_all_filters_normalised(W):
    WT, wavelet = W.WT, W.wavelet
    names = (["haar", "beyl", "vaid"] + [f"db{i}" for i in range(1, 11)] + [f"coif{i}" for i in (2, 4, 6, 8)]
             + [f"sym{i}" for i in range(4, 11)] + [f"batt{i}" for i in (2, 4, 6)])
    for nm in names:
        f = wavelet(getattr(WT, nm))
        assert f.name == nm
        assert abs(np.linalg.norm(f.qmf) - 1) < 1e-12
        assert abs(f.qmf.sum() - np.sqrt(2)) < (1e-3 if nm.startswith("batt") else 1e-8), nm
    assert len(wavelet(WT.db1)) == 2 and len(wavelet(WT.db10)) == 20 and len(wavelet(WT.batt6)) == 59
    assert np.allclose(wavelet(WT.db1).qmf, wavelet(WT.haar).qmf)


def test_daubechies_values(W):
    db4 = [0.23037781330889653, 0.7148465705529158, 0.630880767929859, -0.02798376941686001,
           -0.18703481171909314, 0.03084138183556073, 0.03288301166688519, -0.01059740178506904]
    db2 = [0.4829629131445342, 0.8365163037378079, 0.2241438680420133, -0.12940952255126045]
    assert np.abs(W.wavelet(W.WT.db4).qmf - db4).max() < 1e-13
    assert np.abs(W.wavelet(W.WT.db2).qmf - db2).max() < 1e-13


def test_daubechies_taps_are_pinned_bit_for_bit(W):
    """daubechies(N) follows the reference's order of operations (vieta on the complex zeros, `rmul!(HH, 1/norm(HH))` on the complex
    coefficients, `real` last: wt_main.jl:271-320); the taps of this release are pinned as hex floats so that a rewrite that moves
    one by an ulp (and with it every db-N result) is seen.  The fixture is the implementation's own output -- a drift detector."""
    import json
    import os
    from wavelets_jl_amd import wt
    pin = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "daubechies_taps_pinned.json")))["taps"]
    for N in range(1, 11):
        q = wt.daubechies(N)
        assert [float(v).hex() for v in q] == pin[str(N)], N
        assert np.array_equal(W.wavelet(getattr(W.WT, f"db{N}")).qmf, q / np.linalg.norm(q))      # the constructor renormalises (wt_main.jl:153)


def test_qmf_pairs(W):
    f = W.wavelet(W.WT.db2)
    sc, dc = W.WT.makereverseqmfpair(f, True)
    assert np.array_equal(sc, f.qmf[::-1]) and np.array_equal(dc, f.qmf * [1, -1, 1, -1])
    sc, dc = W.WT.makereverseqmfpair(f, False)
    assert np.array_equal(sc, f.qmf) and np.array_equal(dc, (f.qmf * [1, -1, 1, -1])[::-1])
    s32, _ = W.WT.makereverseqmfpair(f, True, np.float32)
    assert s32.dtype == np.float32


def test_util(W):
    assert W.maxtransformlevels(1) == 0 and W.maxtransformlevels(40) == 3 and W.maxtransformlevels(2 ** 20) == 20
    assert W.maxtransformlevels(np.zeros((8, 32))) == 3
    assert W.sufficientpoweroftwo(24, 3) and not W.sufficientpoweroftwo(24, 4)
    assert W.detailindex(64, 2, 1) == 17 and W.detailn(64, 2) == 16
    assert list(W.detailrange(64, 1)) == list(range(33, 65))
    t = W.maketree(16, 2, "full")
    assert t.tolist() == [1, 1, 1] + [0] * 12 and W.isvalidtree(np.zeros(16), t)
    t = W.maketree(16, 3, "dwt")
    assert np.nonzero(t)[0].tolist() == [0, 1, 3]
    bad = np.zeros(15, dtype=np.uint8)
    bad[1] = 1
    assert not W.isvalidtree(np.zeros(16), bad)
    assert W.iscube(np.zeros((4, 4, 4))) and not W.iscube(np.zeros((4, 8)))


def test_scheme_flatten(W):
    g = W.wavelet(W.WT.db2, W.WT.Lifting)
    iu, nc, sh, cf = g.flatten()
    assert iu.tolist() == [0, 1, 0] and nc.tolist() == [1, 2, 1] and sh.tolist() == [0, 1, -1]
    assert cf.tolist() == [-1.7320508075688772, -0.0669872981077807, 0.4330127018922193, 1.0]


def test_no_cpu_path(W):
    """The product must fail loudly without a device -- never fall back to the CPU."""
    import torch
    x = np.zeros(16, dtype=np.float32)
    with pytest.raises(TypeError):
        W.dwt(x, W.wavelet(W.WT.db2))
    with pytest.raises(W.HIPError):
        W.dwt(torch.zeros(16), W.wavelet(W.WT.db2))
    with pytest.raises(W.HIPError):
        W.dwt_(torch.zeros(16), torch.zeros(16), W.wavelet(W.WT.db2))


def test_product_does_not_import_oracle():
    """Nothing under wavelets.jl_amd/ (nor the C ABI sources) may reference the oracle."""
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavelets.jl_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"wl_oracle|libwl_oracle|import oracle|from oracle|wlo_", txt), (dp, f)


def test_bench_refuses_to_run_without_gpu():
    """bench.py measures the HIP path only; on a GPU-less host it must stop, not fall back."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU path" in (p.stderr + p.stdout)


def test_threshold_and_modwt_host_logic(W):
    """Host-side mirror of Threshold / MODWT: types, defaults, argument contract -- and no CPU path."""
    import math
    import torch
    assert isinstance(W.DEFAULT_TH, W.HardTH) and W.DEFAULT_WAVELET.name == "sym5" and len(W.DEFAULT_WAVELET.qmf) == 10
    vs = W.VisuShrink(256)
    assert isinstance(vs.th, W.HardTH) and vs.t == math.sqrt(2 * math.log(256))         # denoising.jl:14-17
    vs2 = W.VisuShrink(W.SoftTH(), 1.5)
    assert isinstance(vs2.th, W.SoftTH) and vs2.t == 1.5
    # nspin2circ: CartesianIndices order, first dimension fastest (denoising.jl:112-121)
    assert W.nspin2circ(8, 1) == [0] and W.nspin2circ(8, 8) == [7]
    assert [W.nspin2circ((2, 3), i) for i in range(1, 7)] == [[0, 0], [1, 0], [0, 1], [1, 1], [0, 2], [1, 2]]
    assert W.maxmodwttransformlevels(129) == 7 and W.maxmodwttransformlevels(128) == 7 and W.maxmodwttransformlevels(127) == 6
    codes = [t.code for t in (W.HardTH(), W.SoftTH(), W.SemiSoftTH(), W.SteinTH(), W.PosTH(), W.NegTH())]
    assert codes == [0, 1, 2, 3, 4, 5]                                                   # enum wl_thtype
    x = torch.zeros(16)
    for call in (lambda: W.threshold(x, W.HardTH(), 1.0), lambda: W.denoise(x), lambda: W.modwt(x, W.wavelet(W.WT.db2)),
                 lambda: W.mad_(x), lambda: W.noisest(x), lambda: W.imodwt(torch.zeros(16, 3), W.wavelet(W.WT.db2))):
        with pytest.raises(W.HIPError):
            call()
    with pytest.raises(TypeError):
        W.modwt(x, W.wavelet(W.WT.cdf97, W.WT.Lifting))                                  # MethodError in the reference


def test_util_helpers(W):
    """src/Util mirror: dyadic indexing, up/downsampling, wcount, the test signals (test/util.jl style checks)."""
    assert W.dyadicdetailindex(3, 2) == 10 and list(W.dyadicdetailrange(2)) == [5, 6, 7, 8] and list(W.dyadicscalingrange(2)) == [1, 2, 3, 4]
    assert W.dyadicdetailn(5) == 32 and W.maxdyadiclevel(64) == 5 and W.tl2dyadiclevel(64, 2) == 4 and W.dyadiclevel2tl(64, 4) == 2
    assert np.array_equal(W.mirror([1.0, 2.0, 3.0, 4.0]), [1.0, -2.0, 3.0, -4.0])
    x = np.array([1.0, 2.0, 3.0])
    assert np.array_equal(W.upsample(x), [1, 0, 2, 0, 3, 0]) and np.array_equal(W.upsample(x, 1), [0, 1, 0, 2, 0, 3])
    y = np.arange(1.0, 9.0)
    assert np.array_equal(W.downsample(y), [1, 3, 5, 7]) and np.array_equal(W.downsample(y, 1), [2, 4, 6, 8])
    assert np.array_equal(W.downsample(W.upsample(x)), x)
    v = np.array([5.0, -0.1, 0.3, -2.0, 0.0, 0.2, 1.0, -1.0])
    assert W.wcount(v) == 8 and W.wcount(v, 0.25) == 5 and W.wcount(v, 0.25, level=1) == 4 and W.wcount(v.reshape(2, 4), 1.0) == 4
    for ft in ("Blocks", "Bumps", "HeaviSine", "Doppler"):
        f = W.testfunction(256, ft)
        assert f.shape == (256,) and f.dtype == np.float64 and np.all(np.isfinite(f))
    assert W.testfunction(4, "Doppler")[0] == 0.0 and abs(W.testfunction(1024, "Blocks").max() - 5.2) < 1e-12
    with pytest.raises(ValueError):
        W.testfunction(8, "nope")


def test_set_arithmetic_leaves_the_mode_alone_when_the_target_library_is_missing(monkeypatch, tmp_path):
    """set_arithmetic() binds the target library before it retires anything (round-5 review): a missing or stale fused library
    raises and the process stays in the mode it was in."""
    import wavelets_jl_amd as W
    from wavelets_jl_amd import _lib
    assert W.get_arithmetic() == "exact"
    monkeypatch.setitem(_lib.LIB_PATHS, "fused", str(tmp_path / "libwavelets_mi355x_fma.so"))
    monkeypatch.delitem(_lib._libs, "fused", raising=False)
    with pytest.raises(_lib.WaveletsLibraryError):
        W.set_arithmetic("fused")
    assert W.get_arithmetic() == "exact"
    with pytest.raises(_lib.WaveletsLibraryError):
        with W.arithmetic("fused"):
            pass
    assert W.get_arithmetic() == "exact"


def test_makefile_keeps_the_arithmetic_contract_under_a_cxxflags_override():
    """`make FMA=1 CXXFLAGS=...` must still compile with -DWL_FMA -ffp-contract=fast, and the default build with -ffp-contract=off:
    the contract flags are appended outside the overridable variable."""
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavelets.jl_amd", "csrc")
    fma = subprocess.run(["make", "-n", "-B", "-C", csrc, "FMA=1", "CXXFLAGS=-O1"], capture_output=True, text=True, check=True).stdout
    lines = [l for l in fma.splitlines() if " -c " in l]
    assert lines and all("-DWL_FMA" in l and "-ffp-contract=fast" in l and " -O1 " in l for l in lines)
    exact = subprocess.run(["make", "-n", "-B", "-C", csrc, "CXXFLAGS=-O1"], capture_output=True, text=True, check=True).stdout
    lines = [l for l in exact.splitlines() if " -c " in l]
    assert lines and all("-ffp-contract=off" in l and "WL_FMA" not in l for l in lines)
