"""Scratch: host-side enqueue cost per call (no synchronisation inside the timed region)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
def host_cost(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return dt / reps * 1e6
db4 = W.wavelet(W.WT.db4); cdf = W.wavelet(W.WT.cdf97, W.WT.Lifting)
x1 = torch.randn(1 << 14, dtype=torch.float32, device="cuda"); y1 = W.similar(x1)      # tiny: the GPU never limits
for label, fn in (("dwt_oop_ filter 1-D 2^14 L=14", lambda: W.dwt_oop_(y1, x1, db4, 14)),
                  ("dwt_oop_ lifting 1-D 2^14 L=14", lambda: W.dwt_oop_(y1, x1, cdf, 14)),
                  ("threshold_", lambda: W.threshold_(y1, W.HardTH(), 0.1))):
    print(f"{label}: {host_cost(fn):.1f} us host per call")
x = torch.randn(1 << 24, dtype=torch.float32, device="cuda"); y = W.similar(x)
for label, fn in (("dwt_oop_ filter 2^24 L=24", lambda: W.dwt_oop_(y, x, db4, 24)), ("dwt_oop_ lifting 2^24 L=24", lambda: W.dwt_oop_(y, x, cdf, 24)),
                  ("idwt_oop_ lifting 2^24 L=24", lambda: W.idwt_oop_(x, y, cdf, 24))):
    print(f"{label}: {host_cost(fn, 50):.1f} us per call (enqueue-bound or device-bound, whichever is larger)")
