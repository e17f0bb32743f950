"""Pins the CPU oracle against every golden vector the reference's own tests hold for the
transform path (test/transforms.jl:2-55; data files test/data/*.txt committed verbatim under
tests/golden/), with the reference's own tolerances."""
import numpy as np
import pytest

from conftest import golden, golden_cases, make_filter


@pytest.mark.parametrize("fam,num,cls,vm", golden_cases())
def test_accuracy_1d_2d(oracle, W, fam, num, cls, vm):
    data = golden("filter1d_data.txt")
    data2 = golden("filter2d_data.txt")
    assert data.shape == (64,) and data2.shape == (8, 8)
    stderr1 = 1e-9 * np.sqrt(data.size)          # test/transforms.jl:15-16
    stderr2 = 1e-9 * np.sqrt(data2.size)
    ye = golden(f"filter1d_{fam}{num}.txt")
    ye2 = golden(f"filter2d_{fam}{num}.txt")
    wt = make_filter(W, cls, vm)
    y = oracle.dwt_filter(data, wt.qmf)          # full depth, L = 6
    y2 = oracle.dwt_filter(data2, wt.qmf)        # L = 3
    assert np.linalg.norm(y - ye) <= stderr1
    assert np.linalg.norm(y2 - ye2) <= stderr2
    if fam != "Battle":                          # Battle tables are not orthogonal (test/transforms.jl:38-44)
        tol = 1e-9 if not (fam == "Coiflet" and vm == 10) else 1e-6
        assert abs(np.linalg.norm(data) - np.linalg.norm(y)) < tol
        assert abs(np.linalg.norm(data2) - np.linalg.norm(y2)) < tol
        assert np.linalg.norm(oracle.dwt_filter(y, wt.qmf, fw=False) - data) <= stderr1 * 100
        assert np.linalg.norm(oracle.dwt_filter(y2, wt.qmf, fw=False) - data2) <= stderr2 * 100


def test_accuracy_nonsquare(oracle, W):
    data2 = golden("filter2d_nonsquare_data.txt")          # 4 x 8, test/transforms.jl:49-55
    ye2 = golden("filter2d_nonsquare_Haar0.txt")
    y2 = oracle.dwt_filter(data2, W.wavelet(W.WT.haar).qmf, L=1)
    assert np.linalg.norm(y2 - ye2) <= 1e-9 * np.sqrt(ye2.size)


def test_golden_in_float32(oracle, W):
    """Float32 run of the same vectors stays within Float32 round-off of the golden values."""
    data = golden("filter1d_data.txt").astype(np.float32)
    for cls, vm, fam, num in (("Daubechies", 4, "Daubechies", 8), ("Daubechies", 2, "Daubechies", 4), ("Haar", 0, "Haar", 0)):
        wt = make_filter(W, cls, vm)
        y = oracle.dwt_filter(data, wt.qmf)
        assert y.dtype == np.float32
        ye = golden(f"filter1d_{fam}{num}.txt")
        assert np.linalg.norm(y - ye) / np.linalg.norm(ye) < 2e-6
