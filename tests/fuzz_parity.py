"""Differential fuzzing of the HIP path against the CPU oracle: random shapes (with odd factors), filters, lifting
schemes, depths, element types and entry points; every result must be bit-identical.  Usage:
    python tests/fuzz_parity.py [ncases] [seed]        (needs an MI355X; test infrastructure, not product code)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
import wavelets_jl_amd as W

FILTERS = ["haar", "db2", "db3", "db4", "db5", "db6", "db7", "db8", "db9", "db10", "sym4", "sym5", "sym6", "sym8", "sym10",
           "coif2", "coif4", "coif6", "coif8", "batt2", "batt4", "batt6", "vaid", "beyl"]
SCHEMES = ["haar", "db2", "cdf97"]


def rand_len(r, lo, hi):
    """a length m * 2^k within [lo, hi], m odd and small"""
    khi = int(np.log2(hi))
    for _ in range(100):
        m = int(r.choice([1, 1, 1, 1, 3, 5, 7, 9, 15]))
        k = int(r.integers(max(1, khi - 9), khi + 1)) if r.random() < 0.7 else int(r.integers(1, khi + 1))
        n = m << k
        if lo <= n <= hi:
            return n
    return 1 << khi


def one_case(r, verbose=False):
    nd = int(r.choice([1, 1, 2, 2, 2, 3]))
    dtype = np.float32 if r.random() < 0.6 else np.float64
    lifting = r.random() < 0.3
    if nd == 1:
        shape = (rand_len(r, 2, 1 << 21),)
    elif nd == 2:
        if lifting:
            n = rand_len(r, 2, 2048)
            shape = (n, n)
        else:
            shape = (rand_len(r, 2, 4096), rand_len(r, 2, 4096))
            while shape[0] * shape[1] > (1 << 22):
                shape = (shape[0], max(2, shape[1] // 2))
    else:
        n = rand_len(r, 2, 128)
        shape = (n, n, n) if lifting else (n, rand_len(r, 2, 128), rand_len(r, 2, 64))
    x = r.standard_normal(shape).astype(dtype)
    Lmax = W.maxtransformlevels(x)
    L = Lmax if r.random() < 0.5 else int(r.integers(min(1, Lmax), Lmax + 1))
    xd = W.to_device(x)
    if nd == 1 and shape[0] >= 4 and r.random() < 0.2:          # packet transform over a random valid tree (round 5: split masks)
        n = shape[0]
        Lm = W.maxtransformlevels(n)
        depth = int(r.integers(1, min(Lm, 12) + 1)) if Lm >= 1 else 0
        tree = np.zeros(2 ** Lm - 1, dtype=np.uint8)
        if depth >= 1 and len(tree):
            tree[0] = 1
            p = float(r.choice([0.35, 0.6, 0.85, 1.0]))
            for i in range(1, min(len(tree), 2 ** depth - 1)):
                tree[i] = 1 if (tree[(i + 1) // 2 - 1] and r.random() < p) else 0
        if lifting:
            sch = W.wavelet(getattr(W.WT, str(r.choice(SCHEMES))), W.WT.Lifting)
            ye = oracle.wpt_lifting(x, sch, tree)
            y = W.to_host(W.wpt(xd, sch, tree)); kf = W.last_kernel()
            xr = W.to_host(W.iwpt(W.to_device(ye), sch, tree)); ki = W.last_kernel()
            xe = oracle.wpt_lifting(ye, sch, tree, fw=False)
            tag = f"wpt lifting {sch.name} {dtype.__name__} n={n} depth={depth} [{kf} | {ki}]"
        else:
            name = str(r.choice(FILTERS))
            wt = W.wavelet(getattr(W.WT, name))
            ye = oracle.wpt_filter(x, wt.qmf, tree)
            y = W.to_host(W.wpt(xd, wt, tree)); kf = W.last_kernel()
            xr = W.to_host(W.iwpt(W.to_device(ye), wt, tree)); ki = W.last_kernel()
            xe = oracle.wpt_filter(ye, wt.qmf, tree, fw=False)
            tag = f"wpt {name} {dtype.__name__} n={n} depth={depth} [{kf} | {ki}]"
        if verbose:
            print(tag)
        assert np.array_equal(y, ye), "FORWARD " + tag
        assert np.array_equal(xr, xe), "INVERSE " + tag
        return (kf, ki)
    if nd == 2 and not lifting and r.random() < 0.25:          # batched columns: every column its own 1-D transform
        name = str(r.choice(FILTERS))
        wt = W.wavelet(getattr(W.WT, name))
        Lc = int(r.integers(1, W.maxtransformlevels(shape[0]) + 1)) if W.maxtransformlevels(shape[0]) >= 1 else 0
        ye = oracle.dwtc_filter(x, wt.qmf, Lc)
        y = W.to_host(W.dwtc(xd, wt, Lc)); kf = W.last_kernel()
        xr = W.to_host(W.idwtc(W.to_device(ye), wt, Lc)); ki = W.last_kernel()
        tag = f"dwtc {name} {dtype.__name__} shape={shape} L={Lc} [{kf} | {ki}]"
        if verbose:
            print(tag)
        assert np.array_equal(y, ye), "FORWARD " + tag
        assert np.array_equal(xr, oracle.dwtc_filter(ye, wt.qmf, Lc, fw=False)), "INVERSE " + tag
        return kf, ki
    if lifting:
        name = str(r.choice(SCHEMES))
        wt = W.wavelet(getattr(W.WT, name), W.WT.Lifting)
        ye = oracle.dwt_lifting(x, wt, L)
        xe = oracle.dwt_lifting(ye, wt, L, fw=False)
    else:
        name = str(r.choice(FILTERS))
        wt = W.wavelet(getattr(W.WT, name))
        ye = oracle.dwt_filter(x, wt.qmf, L)
        xe = oracle.dwt_filter(ye, wt.qmf, L, fw=False)
    # a third of the cases pull the size thresholds of the fused-pair / single-pass kernels down to the fuzzed sizes and vary
    # their strip / chunk shapes (the default dispatch only uses them from 4096^2 upwards)
    knobs = {}
    if not lifting and nd == 2 and r.random() < 0.35:
        knobs = {"WL_LDS_PAIR_MIN": 0, "WL_LDS_PAIR_MIN64": 0, "WL_LONG2D_MIN_ROWS": 256, "WL_TILE": int(r.integers(0, 2)),
                 "WL_PAIR_W": int(r.choice([2, 4])), "WL_PAIR_W64": int(r.choice([2, 4])), "WL_TJ2": int(r.choice([32, 64, 128])),
                 "WL_LONG_W": int(r.choice([0, 1, 2, 4])), "WL_LONG_TJ": int(r.choice([16, 22, 64, 128])),
                 "WL_INV_PAIR_MIN": int(r.choice([0, 1 << 24])), "WL_TILE_INV": int(r.integers(0, 2)), "WL_TILEB": int(r.integers(0, 2)), "WL_TILEB_MIN": int(r.choice([0, 1 << 21])), "WL_TILEB_MAX": int(r.choice([2048, 4096]))}
    if lifting and nd == 2 and r.random() < 0.35:              # the marching level kernels instead of the 64 x 64 tiles
        knobs = {"WL_LIFT_TILE": 0}
    for k, v in knobs.items():
        W.set_option(k, v)
    try:
        y = W.to_host(W.dwt(xd, wt, L)); kf = W.last_kernel()
        xr = W.to_host(W.idwt(W.to_device(ye), wt, L)); ki = W.last_kernel()
    finally:
        if knobs:
            W.clear_options()
    tag = f"{'lifting' if lifting else 'filter'} {name} {dtype.__name__} shape={shape} L={L} {knobs if knobs else ''} [{kf} | {ki}]"
    if verbose:
        print(tag)
    assert np.array_equal(y, ye), "FORWARD " + tag
    assert np.array_equal(xr, xe), "INVERSE " + tag
    return kf, ki


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    r = np.random.default_rng(seed)
    seen = {}
    for i in range(n):
        kf, ki = one_case(r, verbose=("-v" in sys.argv))
        seen[kf] = seen.get(kf, 0) + 1
        seen[ki] = seen.get(ki, 0) + 1
    print(f"{n} cases bit-identical (seed {seed}); kernels hit:", dict(sorted(seen.items())))


if __name__ == "__main__":
    main()
