"""Scratch: long filters (F > 10) go through the generic kernels: how slow?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name in ("db4", "sym5", "db6", "sym8", "db10", "coif6", "batt4"):
    wt = W.wavelet(getattr(W.WT, name))
    x = torch.randn(8192, 8192, dtype=torch.float32, device="cuda").t(); y = W.similar(x)
    t2 = timeit(lambda: W.dwt_oop_(y, x, wt, 13)); k2 = W.last_kernel()
    t2i = timeit(lambda: W.idwt_oop_(x, y, wt, 13))
    v = torch.randn(1 << 24, dtype=torch.float32, device="cuda"); yv = W.similar(v)
    t1 = timeit(lambda: W.dwt_oop_(yv, v, wt, 24)); k1 = W.last_kernel()
    t1i = timeit(lambda: W.idwt_oop_(v, yv, wt, 24))
    print(f"{name} F={len(wt.qmf)}: 2-D 8192^2 dwt {t2:.0f} us [{k2}] idwt {t2i:.0f} us | 1-D 2^24 dwt {t1:.0f} us [{k1}] idwt {t1i:.0f} us")
