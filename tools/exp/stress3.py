import sys, numpy as np
sys.path.insert(0, '.')
import oracle, wavelets_jl_amd as W
import torch
W._lib.load()
rng = np.random.default_rng(1)
shape = (1024, 2048); L = 2
x = rng.standard_normal(shape).astype(np.float32)
xd = W.to_device(x)
filters = ("db4", "haar", "db2", "db3", "sym5")
exp = {f: oracle.dwt_filter(x, W.wavelet(getattr(W.WT, f)).qmf, L) for f in filters}
W.clear_options()
for k, v in {"WL_LDS_PAIR_MIN": 0, "WL_PAIR_W": 2, "WL_TJ2": 128, "WL_PAIR_WG_PER_CU": 0, "WL_M2D_MAX": 128, "WL_TILE": 0}.items(): W.set_option(k, v)
nfail = 0
for it in range(120):
    for f in filters:
        wt = W.wavelet(getattr(W.WT, f))
        y = W.to_host(W.dwt(xd, wt, L))
        ye = exp[f]
        if not np.array_equal(y, ye):
            nfail += 1
            bad = (y != ye)
            m, n = shape
            q = {"LL2": bad[:m//4, :n//4].sum(), "ds2": bad[m//4:m//2, :n//4].sum(), "sd2": bad[:m//4, n//4:n//2].sum(), "dd2": bad[m//4:m//2, n//4:n//2].sum(),
                 "ds1": bad[m//2:, :n//2].sum(), "sd1": bad[:m//2, n//2:].sum(), "dd1": bad[m//2:, n//2:].sum()}
            cols = np.unique(np.argwhere(bad)[:, 1]); rows = np.unique(np.argwhere(bad)[:, 0])
            print(it, f, "bad", int(bad.sum()), {k: int(v) for k, v in q.items()}, "ncols", len(cols), cols[:20].tolist(), "nrows", len(rows), rows[:10].tolist(), rows[-5:].tolist(), flush=True)
            if nfail > 6: sys.exit(0)
print("failures", nfail)
