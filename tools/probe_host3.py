import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
cdf = W.wavelet(W.WT.cdf97, W.WT.Lifting); db4 = W.wavelet(W.WT.db4)
x = torch.randn(1 << 24, dtype=torch.float32, device="cuda"); y = W.similar(x)
for wt, name in ((db4, "filter"), (cdf, "lifting"), (db4, "filter again"), (cdf, "lifting again")):
    for _ in range(20): W.dwt_oop_(y, x, wt, 24)
    torch.cuda.synchronize()
    ts = []
    t00 = time.perf_counter()
    for i in range(500):
        t = time.perf_counter(); W.dwt_oop_(y, x, wt, 24); ts.append((time.perf_counter() - t) * 1e6)
    t_enq = time.perf_counter() - t00
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t00
    big = [(i, round(v)) for i, v in enumerate(ts) if v > 200]
    print(name, f"enqueue loop {t_enq*1e3:.1f} ms, until sync {t_all*1e3:.1f} ms, median call {sorted(ts)[250]:.0f} us, calls > 200 us: {big[:12]} (n={len(big)})")
