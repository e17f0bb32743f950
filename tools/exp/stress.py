import sys, numpy as np
sys.path.insert(0, '.')
import oracle, wavelets_jl_amd as W
import torch
W._lib.load()
rng = np.random.default_rng(1)
cases = [("sym5", (512, 1056), 4, 2, 128), ("sym5", (512, 1056), 2, 2, 128), ("db4", (512, 1056), 2, 2, 128), ("sym5", (1024, 2048), 2, 4, 64), ("sym5", (2048, 512), 2, 2, 32),
         ("db4", (2048, 2048), 2, 2, 128), ("sym5", (2048, 2048), 2, 2, 128)]
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for fname, shape, L, w, tj in cases:
    wt = W.wavelet(getattr(W.WT, fname))
    x = rng.standard_normal(shape).astype(np.float32)
    ye = oracle.dwt_filter(x, wt.qmf, L)
    xd = W.to_device(x)
    W.clear_options()
    for k, v in {"WL_LDS_PAIR_MIN": 0, "WL_PAIR_W": w, "WL_TJ2": tj, "WL_PAIR_WG_PER_CU": 0, "WL_M2D_MAX": 128, "WL_TILE": 0}.items(): W.set_option(k, v)
    nbad = 0
    for r in range(nrep):
        y = W.to_host(W.dwt(xd, wt, L))
        bad = np.argwhere(y != ye)
        if len(bad):
            nbad += 1
            if nbad <= 3:
                print("  BAD", fname, shape, L, w, tj, "rep", r, len(bad), "rows", bad[:,0].min(), bad[:,0].max(), "cols", bad[:,1].min(), bad[:,1].max(),
                      "uniq rows", len(np.unique(bad[:,0])), "uniq cols", np.unique(bad[:,1])[:12].tolist())
    print(fname, shape, L, w, tj, W.last_kernel(), "failures", nbad, "/", nrep, flush=True)
