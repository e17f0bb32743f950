"""Scratch: full-depth 2-D dwt timings at several sizes (device time, median of 30)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
wt = W.wavelet(W.WT.db4)
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
out = []
for n in (8192, 4096, 2048, 1024, 512):
    x = torch.randn(n, n, dtype=torch.float32, device="cuda").t()
    y = W.similar(x)
    L = W.maxtransformlevels(x)
    out.append(f"{n}:{timeit(lambda: W.dwt_oop_(y, x, wt, L)):.1f}")
print("WAVES_MIN", os.environ.get("WL_WAVES_MIN"), "WPC", os.environ.get("WL_WAVES_PER_CU"), " ".join(out))
