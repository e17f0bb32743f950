"""Scratch measurement: device time of small transforms that run entirely in k_tail_fwd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
wt = W.wavelet(W.WT.db4)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for n in (128, 64, 32, 16):
    x = torch.randn(n, n, dtype=torch.float32, device="cuda").t()
    y = W.similar(x)
    Lm = W.maxtransformlevels(x)
    for L in (1, 2, 3, Lm):
        us = timeit(lambda: W.dwt_(y, x, wt, L))
        print(f"n={n:4d} L={L:2d}  {us:7.2f} us/call  kernel={W.last_kernel()}")

# host-side enqueue cost of one full C3 call (no synchronisation inside the loop)
import time
x = torch.randn(8192, 8192, dtype=torch.float32, device="cuda").t()
y = W.similar(x)
W.reserve_workspace(x, 13)
for _ in range(5): W.dwt_(y, x, wt, 13)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): W.dwt_(y, x, wt, 13)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"C3 enqueue {1e6*(t1-t0)/50:.1f} us/call host, total {1e6*(t2-t0)/50:.1f} us/call")
