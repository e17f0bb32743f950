"""Scratch: 2-D lifting at small sizes (LDS tail kernel) and per-level increments."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavelets_jl_amd as W
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name in ("cdf97", "haar"):
    wt = W.wavelet(getattr(W.WT, name), W.WT.Lifting)
    out = []
    for n in (16, 32, 64, 128, 256, 512, 1024):
        x = torch.randn(n, n, dtype=torch.float32, device="cuda").t(); y = W.similar(x)
        L = W.maxtransformlevels(x)
        out.append(f"{n}:{timeit(lambda: W.dwt_oop_(y, x, wt, L)):.0f}/{timeit(lambda: W.idwt_oop_(x, y, wt, L)):.0f}")
    print(name, "full-depth dwt/idwt us:", " ".join(out))
