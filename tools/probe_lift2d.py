"""Scratch: 2-D lifting timings."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavelets_jl_amd as W
def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name in ("cdf97", "db2", "haar"):
    wt = W.wavelet(getattr(W.WT, name), W.WT.Lifting)
    for n in (1024, 4096, 8192):
        x = torch.randn(n, n, dtype=torch.float32, device="cuda").t()
        y = W.similar(x)
        L = W.maxtransformlevels(x)
        t1 = timeit(lambda: W.dwt_oop_(y, x, wt, 1))
        t = timeit(lambda: W.dwt_oop_(y, x, wt, L)); k = W.last_kernel()
        ti = timeit(lambda: W.idwt_oop_(x, y, wt, L))
        print(f"{name} {n}^2 f32: dwt L=1 {t1:.0f} us, L={L} {t:.0f} us, idwt {ti:.0f} us  [{k}]")

wt = W.wavelet(W.WT.cdf97, W.WT.Lifting)
for n in (128, 256, 512):
    x = torch.randn(n, n, n, dtype=torch.float32, device="cuda").permute(2, 1, 0)
    y = W.similar(x)
    L = W.maxtransformlevels(x)
    t = timeit(lambda: W.dwt_oop_(y, x, wt, L)); k = W.last_kernel()
    ti = timeit(lambda: W.idwt_oop_(x, y, wt, L))
    print(f"cdf97 {n}^3 f32 L={L}: dwt {t:.0f} us, idwt {ti:.0f} us  [{k}]")
