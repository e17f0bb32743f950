import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def W():
    import wavelets_jl_amd as w
    return w


@pytest.fixture(scope="session")
def gpu(W):
    """A HIP device, or a hard failure: -m gpu tests must never pass on a fallback."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: the gpu-marked tests need a real MI355X")
    W._lib.load()
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True)
def _reset_library_options(request):
    """Options set with W.set_option() inside a test never leak into the next one."""
    yield
    if "W" in request.fixturenames:
        request.getfixturevalue("W").clear_options()


def golden(name):
    return np.loadtxt(os.path.join(GOLDEN, name))


# The 27 filters of the reference's accuracy test (test/transforms.jl:3-6):
# (golden-file family name, file numbers, WT class name, vanishing-moment numbers)
GOLDEN_FAMILIES = [
    ("Daubechies", list(range(4, 21, 2)), "Daubechies", list(range(2, 11))),
    ("Coiflet", [2, 3, 4, 5], "Coiflet", [4, 6, 8, 10]),
    ("Haar", [0], "Haar", [0]),
    ("Symmlet", list(range(4, 11)), "Symlet", list(range(4, 11))),
    ("Battle", [1, 3, 5], "Battle", [2, 4, 6]),
    ("Vaidyanathan", [0], "Vaidyanathan", [0]),
    ("Beylkin", [0], "Beylkin", [0]),
]


def golden_cases():
    out = []
    for fam, nums, cls, vms in GOLDEN_FAMILIES:
        for num, vm in zip(nums, vms):
            out.append((fam, num, cls, vm))
    return out


def make_filter(W, cls, vm):
    c = getattr(W.WT, cls)
    return W.wavelet(c(vm) if vm != 0 else c(), W.WT.Filter)


def rng_array(shape, dtype, seed):
    r = np.random.default_rng(seed)
    return r.standard_normal(shape).astype(dtype)
