"""Scratch: timings of the 3-D transforms (generic kernels)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavelets_jl_amd as W
wt = W.wavelet(W.WT.db4)
def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for n in (64, 128, 256, 512):
    x = W.to_device(torch.randn(n, n, n).numpy())
    y = W.similar(x)
    L = W.maxtransformlevels(x)
    t = timeit(lambda: W.dwt_(y, x, wt, L))
    ti = timeit(lambda: W.idwt_(x, y, wt, L))
    nb = 2 * x.numel() * 4
    print(f"3-D {n}^3 f32 L={L}: dwt {t:.1f} us ({nb/t/1e3:.0f} GB/s alg), idwt {ti:.1f} us  kernel={W.last_kernel()}")
