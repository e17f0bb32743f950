import sys, numpy as np
sys.path.insert(0, '.')
import oracle, wavelets_jl_amd as W
import torch
W._lib.load()
def rng_array(shape, dtype, seed):
    return np.random.default_rng(seed).standard_normal(shape).astype(dtype)
shapes = (((512, 512), (2, 3)), ((1024, 2048), (2,)), ((2048, 512), (2, 4)), ((1536, 160), (2,)), ((512, 1056), (2,)))
for safe in (0, 1, 0, 1):
    nfail = 0; tot = 0
    for rep in range(int(sys.argv[1])):
      for wmain in (2, 4):
        for tj in (32, 64, 128):
            W.clear_options()
            for k, v in {"WL_LDS_PAIR_MIN": 0, "WL_PAIR_W": wmain, "WL_TJ2": tj, "WL_PAIR_WG_PER_CU": 0, "WL_M2D_MAX": 128, "WL_TILE": 0, "WL_PAIR_SAFEWAIT": safe}.items(): W.set_option(k, v)
            for shape, Ls in shapes:
                x = rng_array(shape, np.float32, sum(shape) + wmain + tj)
                xd = W.to_device(x)
                for fname in ("sym5", "db4"):
                    wt = W.wavelet(getattr(W.WT, fname))
                    for L in Ls:
                        ye = oracle.dwt_filter(x, wt.qmf, L)
                        for r in range(4):
                            y = W.to_host(W.dwt(xd, wt, L))
                            tot += 1
                            if not np.array_equal(y, ye):
                                nfail += 1
                                if nfail < 4: print("   fail", safe, shape, fname, L, wmain, tj, flush=True)
    print("safe", safe, "failures", nfail, "/", tot, flush=True)
