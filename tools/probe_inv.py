"""Scratch: idwt timings (2-D 8192^2 L=13, 1-D 2^24 L=24) for tuning the inverse tail threshold."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavelets_jl_amd as W
wt = W.wavelet(W.WT.db4)
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
x = torch.randn(8192, 8192, dtype=torch.float32, device="cuda").t()
y = W.dwt(x, wt); z = W.similar(x)
t2 = timeit(lambda: W.idwt_(z, y, wt, 13))
t21 = timeit(lambda: W.idwt_(z, y, wt, 1))
print("L=1: %.1f us" % t21, W.last_kernel())
v = torch.randn(1 << 24, dtype=torch.float32, device="cuda"); yv = W.dwt(v, wt); zv = W.similar(v)
t1 = timeit(lambda: W.idwt_(zv, yv, wt))
print("PPL", os.environ.get("WL_INV2D_PPL"), "TP", os.environ.get("WL_INV2D_TP"), "NO", os.environ.get("WL_NO_INV2D"), "idwt2d %.1f us, idwt1d %.1f us" % (t2, t1))
