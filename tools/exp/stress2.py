import sys, numpy as np
sys.path.insert(0, '.')
import oracle, wavelets_jl_amd as W
import torch
W._lib.load()
rng = np.random.default_rng(1)
shapes = (((512, 512), (2, 3, 9)), ((1024, 2048), (2, 5)), ((2048, 512), (2, 4, 9)), ((4096, 64), (2,)), ((1536, 160), (2, 3)),
          ((1024, 96), (2,)), ((512, 1056), (2, 4)))
tot = 0
for w in (2, 4):
    for tj in (32, 64, 128):
        for shape, Ls in shapes:
            x = rng.standard_normal(shape).astype(np.float32)
            xd = W.to_device(x)
            for fname in ("db4", "haar", "db2", "db3", "sym5"):
                wt = W.wavelet(getattr(W.WT, fname))
                for L in Ls:
                    ye = oracle.dwt_filter(x, wt.qmf, L)
                    W.clear_options()
                    for k, v in {"WL_LDS_PAIR_MIN": 0, "WL_PAIR_W": w, "WL_TJ2": tj, "WL_PAIR_WG_PER_CU": 0, "WL_M2D_MAX": 128, "WL_TILE": 0}.items(): W.set_option(k, v)
                    for r in range(3):
                        yd = W.similar(xd); yd.fill_(float("nan"))
                        W.dwt_oop_(yd, xd, wt, L)
                        y = W.to_host(yd)
                        tot += 1
                        bad = np.argwhere(~((y == ye)))
                        if len(bad):
                            print("  BAD", fname, shape, L, w, tj, "rep", r, len(bad), "rows", bad[:,0].min(), bad[:,0].max(), "cols", bad[:,1].min(), bad[:,1].max(),
                                  "uniq rows", len(np.unique(bad[:,0])), "uniq cols", np.unique(bad[:,1])[:12].tolist(), "nan", int(np.isnan(y).sum()), flush=True)
print("total", tot)
