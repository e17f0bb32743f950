"""Scratch: in-place read-modify-write vs out-of-place copy bandwidth at sizes beyond the 256 MB MALL."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavelets_jl_amd as W
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for lg in (24, 26, 28, 30):
    n = 1 << lg
    x = torch.randn(n, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    t_rmw = timeit(lambda: W.threshold_(x, W.NegTH()))          # in place, data-independent traffic
    t_cp = timeit(lambda: y.copy_(x))
    t_mul = timeit(lambda: x.mul_(1.0))
    print(f"n=2^{lg} ({n*4/2**20:.0f} MiB): in-place threshold {8*n/t_rmw/1e3:.0f} GB/s, torch in-place mul_ {8*n/t_mul/1e3:.0f} GB/s, copy_ {8*n/t_cp/1e3:.0f} GB/s")
    del x, y
