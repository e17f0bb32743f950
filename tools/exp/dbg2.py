import sys, numpy as np
sys.path.insert(0, '.')
import oracle, wavelets_jl_amd as W
W._lib.load()
rng = np.random.default_rng(1)
for fname in ("sym5", "db4"):
    wt = W.wavelet(getattr(W.WT, fname))
    for shape, L in (((128, 264), 2), ((128, 264), 1), ((64, 132), 1), ((512, 1056), 4), ((512, 1056), 3), ((512, 1056), 2)):
        for opts in ({}, {"WL_M2D_MAX": 128, "WL_TILE": 0}, {"WL_LDS_PAIR_MIN": 0, "WL_PAIR_W": 2, "WL_TJ2": 128, "WL_PAIR_WG_PER_CU": 0, "WL_M2D_MAX": 128, "WL_TILE": 0}):
            x = rng.standard_normal(shape).astype(np.float32)
            W.clear_options()
            for k, v in opts.items(): W.set_option(k, v)
            y = W.to_host(W.dwt(W.to_device(x), wt, L))
            ye = oracle.dwt_filter(x, wt.qmf, L)
            bad = np.argwhere(y != ye)
            msg = "ok" if len(bad) == 0 else f"BAD {len(bad)} rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()}"
            print(fname, shape, L, len(opts), W.last_kernel(), msg)
