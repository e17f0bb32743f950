import sys, numpy as np
sys.path.insert(0, '.')
import oracle, wavelets_jl_amd as W
W._lib.load()
rng = np.random.default_rng(1)
wt = W.wavelet(W.WT.db4)
for shape in ((512, 512), (1024, 2048), (2048, 512), (4096, 64), (1024, 1024)):
    for w in (4, 2):
        for tj in (32, 64, 128):
            x = rng.standard_normal(shape).astype(np.float32)
            W.clear_options()
            W.set_option("WL_LDS_PAIR_MIN", 0); W.set_option("WL_M2D_MAX", 128); W.set_option("WL_TILE", 0)
            W.set_option("WL_PAIR_W", w); W.set_option("WL_TJ2", tj); W.set_option("WL_PAIR_WG_PER_CU", 0)
            y = W.to_host(W.dwt(W.to_device(x), wt, 2))
            ye = oracle.dwt_filter(x, wt.qmf, 2)
            bad = np.argwhere(y != ye)
            if len(bad) == 0:
                print(shape, w, tj, W.last_kernel(), "ok")
            else:
                r, c = bad[:, 0], bad[:, 1]
                print(shape, w, tj, W.last_kernel(), "BAD", len(bad), "rows", r.min(), r.max(), "cols", c.min(), c.max(),
                      "uniq rows", len(np.unique(r)), "uniq cols", len(np.unique(c)), "first", bad[:6].tolist())
