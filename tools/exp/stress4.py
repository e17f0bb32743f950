import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle, wavelets_jl_amd as W
import torch
W._lib.load()
def rng_array(shape, dtype, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(dtype)
shapes = (((512, 512), (2, 3, 9)), ((1024, 2048), (2, 5)), ((2048, 512), (2, 4, 9)), ((4096, 64), (2,)), ((1536, 160), (2, 3)),
          ((1024, 96), (2,)), ((512, 1056), (2, 4)))
nfail = 0; tot = 0
for rep in range(int(sys.argv[1])):
  for wmain in (2, 4):
    for tj in (32, 64, 128):
        W.clear_options()
        for k, v in {"WL_LDS_PAIR_MIN": 0, "WL_PAIR_W": wmain, "WL_TJ2": tj, "WL_PAIR_WG_PER_CU": 0, "WL_M2D_MAX": 128, "WL_TILE": 0}.items(): W.set_option(k, v)
        for shape, Ls in shapes:
            x = rng_array(shape, np.float32, sum(shape) + wmain + tj)
            for fname in ("db4", "haar", "db2", "db3", "sym5"):
                wt = W.wavelet(getattr(W.WT, fname))
                for L in Ls:
                    yd = W.dwt(W.to_device(x), wt, L)
                    torch.cuda.synchronize()
                    y = W.to_host(yd)
                    ye = oracle.dwt_filter(x, wt.qmf, L)
                    tot += 1
                    if not np.array_equal(y, ye):
                        nfail += 1
                        bad = (y != ye)
                        m, n = shape
                        q = {"LL2": bad[:m//4, :n//4].sum(), "ds2": bad[m//4:m//2, :n//4].sum(), "sd2": bad[:m//4, n//4:n//2].sum(), "dd2": bad[m//4:m//2, n//4:n//2].sum(),
                             "ds1": bad[m//2:, :n//2].sum(), "sd1": bad[:m//2, n//2:].sum(), "dd1": bad[m//2:, n//2:].sum()}
                        cols = np.unique(np.argwhere(bad)[:, 1]); rows = np.unique(np.argwhere(bad)[:, 0])
                        # second try of the same call
                        y2 = W.to_host(W.dwt(W.to_device(x), wt, L)); again = np.array_equal(y2, ye)
                        print(rep, shape, fname, L, wmain, tj, "bad", int(bad.sum()), {k: int(v) for k, v in q.items()}, "ncols", len(cols), cols[:16].tolist(), "nrows", len(rows), rows[:8].tolist(), "retry ok" if again else "retry BAD", flush=True)
print("failures", nfail, "/", tot)
