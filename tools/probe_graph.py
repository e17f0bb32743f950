"""Scratch: does hipGraph replay shrink the launch-bound tails?  (torch.cuda.CUDAGraph == hipGraph on ROCm)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
cases = []
x = torch.randn(8192, 8192, dtype=torch.float32, device="cuda").t(); cases.append(("c3 dwt 8192^2 db4 L=13", x, W.wavelet(W.WT.db4), 13))
x = torch.rand(1 << 20, dtype=torch.float64, device="cuda"); cases.append(("c1 dwt 2^20 f64 db2 L=20", x, W.wavelet(W.WT.db2), 20))
x = torch.randn(1024, 1024, dtype=torch.float32, device="cuda").t(); cases.append(("dwt 1024^2 db4 L=10", x, W.wavelet(W.WT.db4), 10))
x = torch.randn(1 << 24, dtype=torch.float32, device="cuda"); cases.append(("c4 lifting cdf97 2^24 L=24", x, W.wavelet(W.WT.cdf97, W.WT.Lifting), 24))
s = torch.cuda.Stream()
for label, x, wt, L in cases:
    y = W.similar(x)
    t_plain = timeit(lambda: W.dwt_oop_(y, x, wt, L))
    ref = y.clone()
    with torch.cuda.stream(s):
        for _ in range(3): W.dwt_oop_(y, x, wt, L)          # creates the context + workspace of stream s
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    y.zero_()
    with torch.cuda.graph(g, stream=s):
        W.dwt_oop_(y, x, wt, L)
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(y, ref)
    t_graph = timeit(lambda: g.replay())
    print(f"{label}: plain {t_plain:.1f} us, graph replay {t_graph:.1f} us, identical={ok}")
