"""WT -- wavelet classes and transform-type descriptors (host-side mirror of the
reference's `WT` module, src/WT/wt_main.jl).

Only what the transform path needs is mirrored: the wavelet classes (`WT.haar`,
`WT.db1..db10`, `WT.coif2..8`, `WT.sym4..10`, `WT.batt2..6`, `WT.beyl`, `WT.vaid`,
`WT.cdf97`; wt_main.jl:74-128), the transform selectors `WT.Filter` / `WT.Lifting`
(:23-28), the boundary singleton `WT.Periodic` (:33-49), `OrthoFilter` (:139-158),
`GLS` / `LSStep` (:195-238) and `wavelet(...)` (:262-264).  Continuous wavelets are out
of scope.

Tap tables for the tabulated families come from wt_tables.json (data extracted from the
reference by tools/gen_wt_tables.py); Daubechies taps are computed by `daubechies(N)`,
which follows the algorithm of wt_main.jl:271-361 (roots of the truncated binomial series
-> z-domain roots inside the unit circle -> Vieta).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from math import comb
from typing import List, Sequence, Tuple

import numpy as np

_TABLES = None


def _tables():
    global _TABLES
    if _TABLES is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "wt_tables.json")) as f:
            _TABLES = json.load(f)
    return _TABLES


# ---- transform types (wt_main.jl:23-28) -------------------------------------------------
class FilterTransform:
    def __repr__(self):
        return "WT.Filter"


class LiftingTransform:
    def __repr__(self):
        return "WT.Lifting"


Filter = FilterTransform()
Lifting = LiftingTransform()


# ---- boundaries (wt_main.jl:33-49); only PerBoundary is implemented by any transform ----
class WaveletBoundary:
    pass


class PerBoundary(WaveletBoundary):
    def __repr__(self):
        return "WT.Periodic"


class ZPBoundary(WaveletBoundary):
    pass


class NullBoundary(WaveletBoundary):
    pass


class SymBoundary(WaveletBoundary):
    pass


Periodic = PerBoundary()
DEFAULT_BOUNDARY = Periodic
padded = ZPBoundary()
NaivePer = NullBoundary()
SymBound = SymBoundary()


# ---- wavelet classes (wt_main.jl:68-128) --------------------------------------------------
class WaveletClass:
    _name = ""
    _moments = None

    def __repr__(self):
        return "WT." + name(self).replace("/", "")

    def __eq__(self, other):
        return type(self) is type(other) and name(self) == name(other)

    def __hash__(self):
        return hash((type(self).__name__, name(self)))


class OrthoWaveletClass(WaveletClass):
    pass


class BiOrthoWaveletClass(WaveletClass):
    pass


def _single(clsname, base, moments):
    return type(clsname, (OrthoWaveletClass,), {"_name": base, "_moments": moments})


Haar = _single("Haar", "haar", 1)
Beylkin = _single("Beylkin", "beyl", -1)
Vaidyanathan = _single("Vaidyanathan", "vaid", -1)


class _Numbered(OrthoWaveletClass):
    _base = ""

    def __init__(self, N: int):
        self.N = int(N)

    @property
    def _name(self):
        return f"{self._base}{self.N}"

    @property
    def _moments(self):
        return self.N


class Daubechies(_Numbered):
    _base = "db"


class Coiflet(_Numbered):
    _base = "coif"


class Symlet(_Numbered):
    _base = "sym"


class Battle(_Numbered):
    _base = "batt"


class CDF(BiOrthoWaveletClass):
    def __init__(self, N1: int, N2: int):
        self.N1, self.N2 = int(N1), int(N2)

    @property
    def _name(self):
        return f"cdf{self.N1}/{self.N2}"

    @property
    def _moments(self):
        return (self.N1, self.N2)


def name(w) -> str:
    """WT.name"""
    if isinstance(w, (OrthoFilter, GLS)):
        return w.name
    return w._name


def vanishingmoments(w):
    return w._moments


def class_(w) -> str:
    return type(w).__name__


# shortcuts: WT.haar, WT.db2, ... (wt_main.jl:85-88,104-109,122-127)
haar = Haar()
beyl = Beylkin()
vaid = Vaidyanathan()
for _n in range(1, 11):
    globals()[f"db{_n}"] = Daubechies(_n)
for _n in range(2, 9, 2):
    globals()[f"coif{_n}"] = Coiflet(_n)
for _n in range(4, 11):
    globals()[f"sym{_n}"] = Symlet(_n)
for _n in range(2, 7, 2):
    globals()[f"batt{_n}"] = Battle(_n)
cdf97 = CDF(9, 7)
# the filter table also has coif10 (wt_main.jl:392-393) although no shortcut exists
# (the Coiflet range is 2:2:8); Coiflet(10) reaches it, as the reference's tests do
# (test/transforms.jl:6 uses WT.Coiflet{10}).


# ---- Daubechies filters (wt_main.jl:271-361) -----------------------------------------------
def daubechies(N: int) -> np.ndarray:
    """Scaling filter of the Daubechies wavelet with N vanishing moments (2N taps, unit 2-norm), by spectral factorisation --
    the construction the reference uses (wt_main.jl:271-361), so that the taps agree with it to rounding:
    |H(w)|^2 = cos^2N(w/2) P(sin^2(w/2)) with P(y) = sum_{n<N} C(N-1+n, n) y^n; every root y of P gives a reciprocal pair of
    zeros z, 1/z through z + 1/z = 2 - 4y; the minimum-phase filter keeps the zeros inside the unit circle next to the N-fold
    zero at z = -1."""
    if N < 1:
        raise AssertionError("N > 0")
    eps = np.finfo(np.float64).eps
    if N == 1:
        inside = np.empty(0, dtype=np.complex128)
    else:
        series = [comb(N - 1 + n, n) for n in range(N)]                  # P, ascending powers of y
        y = np.roots(np.array(series[::-1], dtype=np.float64)).astype(np.complex128)
        centre, radius = 1 - 2 * y, 2 * np.sqrt(y * y - y)              # z = centre +- radius solves z + 1/z = 2 - 4y
        pairs = np.concatenate([centre + radius, centre - radius])
        inside = pairs[np.abs(pairs) <= 1 + eps]
    zeros = np.concatenate([np.full(N, -1.0 + 0j), inside])
    # monic polynomial with these zeros, highest power first, kept COMPLEX (np.poly would drop the imaginary parts of conjugate
    # pairs before the normalisation); then the reference's own order of operations (wt_main.jl:316-319):
    # rmul!(HH, 1/norm(HH)) on the complex coefficients -- a multiplication by the reciprocal of the complex norm -- and real() last
    c = np.ones(1, dtype=np.complex128)
    for r in zeros:
        c = np.concatenate([c, [0]]) - r * np.concatenate([[0], c])      # vieta (wt_main.jl:346-361): C[i+1] = C[i+1] - R[k] * C[i]
    c = c * (1.0 / np.linalg.norm(c))
    return np.ascontiguousarray(c.real, dtype=np.float64)


# ---- OrthoFilter (wt_main.jl:139-163) --------------------------------------------------------
class OrthoFilter:
    """Wavelet type for discrete orthogonal transforms by filtering."""

    def __init__(self, w_or_qmf, boundary: WaveletBoundary = DEFAULT_BOUNDARY, name_: str = None):
        if isinstance(w_or_qmf, OrthoWaveletClass):
            w = w_or_qmf
            nm = name(w)
            if isinstance(w, Daubechies):
                q = daubechies(vanishingmoments(w))
            else:
                q = _tables()["filters"].get(nm)
                if q is None:
                    raise ValueError("filter not found")          # ArgumentError in the reference
                q = np.asarray(q, dtype=np.float64)
            # "make sure it is normalized in l2-norm" (wt_main.jl:152-153)
            self.qmf = q / np.linalg.norm(q)
            self.name = nm
        elif isinstance(w_or_qmf, WaveletClass):
            raise TypeError(f"no OrthoFilter for wavelet class {w_or_qmf!r} (MethodError in the reference)")
        else:
            self.qmf = np.asarray(w_or_qmf, dtype=np.float64).copy()
            self.name = name_ or "custom"
        self.boundary = boundary

    def __len__(self):
        return len(self.qmf)

    def __repr__(self):
        return f"OrthoFilter({self.name!r}, len={len(self)})"


def qmf(f: OrthoFilter) -> np.ndarray:
    return f.qmf


def scale(f: OrthoFilter, a: float) -> OrthoFilter:
    return OrthoFilter(f.qmf * a, f.boundary, f.name)


def mirror(f: np.ndarray) -> np.ndarray:
    """Util.mirror (util_main.jl:30)"""
    f = np.asarray(f)
    return f * (-1.0) ** np.arange(len(f))


def makereverseqmfpair(f: OrthoFilter, fw: bool = True, T=np.float64) -> Tuple[np.ndarray, np.ndarray]:
    """WT.makereverseqmfpair (wt_main.jl:172-183)"""
    h = np.asarray(f.qmf, dtype=T).copy()
    if fw:
        return h[::-1].copy(), mirror(h).astype(T)
    return h, mirror(h).astype(T)[::-1].copy()


def makeqmfpair(f: OrthoFilter, fw: bool = True, T=np.float64):
    s, d = makereverseqmfpair(f, fw, T)
    return s[::-1].copy(), d[::-1].copy()


# ---- lifting schemes (wt_main.jl:195-238) ------------------------------------------------------
class StepType:
    pass


class PredictStep(StepType):
    def __repr__(self):
        return "WT.Predict"


class UpdateStep(StepType):
    def __repr__(self):
        return "WT.Update"


Predict = PredictStep()
Update = UpdateStep()


@dataclass
class LSStepParam:
    coef: np.ndarray
    shift: int

    def __len__(self):
        return len(self.coef)


@dataclass
class LSStep:
    param: LSStepParam
    steptype: StepType

    def __len__(self):
        return len(self.param)


def make_lsstep(st: StepType, coef: Sequence[float], shift: int) -> LSStep:
    return LSStep(LSStepParam(np.asarray(coef, dtype=np.float64), int(shift)), st)


class GLS:
    """Wavelet type for discrete general (bi)orthogonal transforms by a lifting scheme."""

    def __init__(self, w, boundary: WaveletBoundary = DEFAULT_BOUNDARY):
        if isinstance(w, WaveletClass):
            nm = name(w)
            sd = _tables()["schemes"].get(nm)
            if sd is None:
                raise ValueError("scheme not found")              # ArgumentError in the reference
            self.step: List[LSStep] = [
                make_lsstep(Update if s["type"] == "Update" else Predict, s["coef"], s["shift"]) for s in sd["steps"]
            ]
            self.norm1 = float(sd["norm1"])
            self.norm2 = float(sd["norm2"])
            self.name = nm
        else:
            steps, n1, n2, nm = w
            self.step, self.norm1, self.norm2, self.name = list(steps), float(n1), float(n2), nm
        self.boundary = boundary

    def __repr__(self):
        return f"GLS({self.name!r}, steps={len(self.step)})"

    def flatten(self):
        """(is_update[int32], ncoef[int32], shift[int32], coefs[float64]) in table order -- the
        form the C ABI takes (include/wavelets_mi355x.h: wl_dwt_lifting)."""
        cached = getattr(self, "_flat", None)          # a scheme is immutable: flatten once, not on every transform call
        if cached is not None:
            return cached
        iu = np.array([1 if isinstance(s.steptype, UpdateStep) else 0 for s in self.step], dtype=np.int32)
        nc = np.array([len(s) for s in self.step], dtype=np.int32)
        sh = np.array([s.param.shift for s in self.step], dtype=np.int32)
        cf = np.concatenate([s.param.coef for s in self.step]).astype(np.float64) if self.step else np.zeros(0)
        self._flat = (iu, nc, sh, cf)
        return self._flat


# ---- wavelet(...) (wt_main.jl:262-264) ---------------------------------------------------------
def wavelet(c, *args):
    """wavelet(c[, t=WT.Filter][, boundary=WT.Periodic])"""
    t = Filter
    boundary = DEFAULT_BOUNDARY
    rest = list(args)
    if rest and isinstance(rest[0], (FilterTransform, LiftingTransform)):
        t = rest.pop(0)
    if rest and isinstance(rest[0], WaveletBoundary):
        boundary = rest.pop(0)
    if rest:
        raise TypeError("wavelet: unsupported arguments (MethodError in the reference)")
    if not isinstance(c, WaveletClass):
        raise TypeError("wavelet: first argument must be a wavelet class")
    if isinstance(t, LiftingTransform):
        return GLS(c, boundary)
    if isinstance(c, OrthoWaveletClass):
        return OrthoFilter(c, boundary)
    # wavelet(WT.cdf97) / wavelet(WT.cdf97, WT.Filter) has no method in the reference
    raise TypeError(f"no method wavelet({c!r}, WT.Filter) (MethodError in the reference)")
