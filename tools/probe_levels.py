"""Scratch: time single 2-D levels of various sizes (L=1 calls) to tune chunking."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
wt = W.wavelet(W.WT.db4)
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in ev)
    return d[len(d)//2] * 1e3
out = []
for n in (4096, 2048, 1024, 512, 256):
    x = torch.randn(n, n, dtype=torch.float32, device="cuda").t()
    y = W.similar(x)
    us = timeit(lambda: W.dwt_(y, x, wt, 1))
    out.append(f"{n}:{us:.1f}")
print("wpc", os.environ.get("WL_WAVES_PER_CU"), "tj", os.environ.get("WL_TJ"), " ".join(out))
