"""Scratch: idwt 2-D per-size timings (L = 1 and full depth) to see where the deep levels lose time."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavelets_jl_amd as W
wt = W.wavelet(W.WT.db4)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for k in range(4, 14):
    n = 1 << k
    x = torch.randn(n, n, dtype=torch.float32, device="cuda").t()
    z = W.similar(x)
    t1 = timeit(lambda: W.idwt_(z, x, wt, 1)); k1 = W.last_kernel()
    tf = timeit(lambda: W.idwt_(z, x, wt, k)); kf = W.last_kernel()
    f1 = timeit(lambda: W.dwt_(z, x, wt, 1))
    ff = timeit(lambda: W.dwt_(z, x, wt, k))
    print(f"n=2^{k}: idwt L=1 {t1:.1f} us ({k1}), full {tf:.1f} us ({kf}) | dwt L=1 {f1:.1f}, full {ff:.1f}")
