"""Scratch: timings of the section 8(f) rows 3-4 entry points (modwt, threshold, mad, denoise)."""
import torch, sys, os, time
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
for n, L in ((1 << 20, 20), (1 << 24, 8), (1 << 24, 24)):
    x = torch.randn(n, dtype=torch.float32, device="cuda")
    t = timeit(lambda: W.modwt(x, wt, L))
    w = W.modwt(x, wt, L)
    ti = timeit(lambda: W.imodwt(w, wt))
    print(f"modwt n=2^{n.bit_length()-1} L={L} f32: {t:.0f} us ({12*n*L/t/1e3:.0f} GB/s level-traffic), imodwt {ti:.0f} us")
x = torch.randn(1 << 26, dtype=torch.float32, device="cuda")
for th in (W.HardTH(), W.SoftTH(), W.SteinTH()):
    t = timeit(lambda: W.threshold_(x, th, 0.5))
    print(f"threshold {th} 2^26 f32 (t Float64): {t:.0f} us ({8*x.numel()/t/1e3:.0f} GB/s)")
t = timeit(lambda: W.threshold_(x, W.HardTH(), 1))
print(f"threshold HardTH 2^26 f32 (t Int): {t:.0f} us ({8*x.numel()/t/1e3:.0f} GB/s)")
v = torch.randn(1 << 23, dtype=torch.float32, device="cuda")
t = timeit(lambda: W.median(v)); print(f"median 2^23 f32: {t:.0f} us")
t = timeit(lambda: W.mad_(v.clone())); print(f"mad (incl. clone) 2^23 f32: {t:.0f} us")
t = timeit(lambda: W.threshold(v, W.BiggestTH(), 1000)); print(f"BiggestTH (incl. copy) 2^23 f32: {t:.0f} us")
for n in (1 << 16, 1 << 20, 1 << 24):
    x = torch.randn(n, dtype=torch.float32, device="cuda")
    t = timeit(lambda: W.denoise(x)); print(f"denoise 1-D n=2^{n.bit_length()-1} f32 (sym5, L=6, hard): {t:.0f} us")
x = torch.randn(1 << 16, dtype=torch.float32, device="cuda")
t = timeit(lambda: W.denoise(x, TI=True)); print(f"denoise TI (8 spins) n=2^16: {t:.0f} us")
a = torch.randn(4096, 4096, dtype=torch.float32, device="cuda").t()
t = timeit(lambda: W.denoise(a)); print(f"denoise 2-D 4096^2 f32: {t:.0f} us")
