import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
cdf = W.wavelet(W.WT.cdf97, W.WT.Lifting); db4 = W.wavelet(W.WT.db4)
x = torch.randn(1 << 24, dtype=torch.float32, device="cuda"); y = W.similar(x)
for wt, name in ((db4, "filter"), (cdf, "lifting")):
    for _ in range(5): W.dwt_oop_(y, x, wt, 24)
    torch.cuda.synchronize()
    ts = []
    for i in range(40):
        t = time.perf_counter(); W.dwt_oop_(y, x, wt, 24); ts.append((time.perf_counter() - t) * 1e6)
    torch.cuda.synchronize()
    print(name, "per-call host us:", " ".join(f"{v:.0f}" for v in ts))
