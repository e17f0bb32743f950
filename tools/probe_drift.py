"""Scratch: per-step device time drift of the C3 transform (is the first timed loop slower?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
wt = W.wavelet(W.WT.db4)
x = torch.randn(8192, 8192, dtype=torch.float32, device="cuda").t()
W.reserve_workspace(x, 13)
for rnd in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for i in range(50):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = W.dwt(x, wt, 13); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    d = [a.elapsed_time(b) * 1e3 for a, b in evs]
    print(f"round {rnd}: wall {wall:.1f} us/step; per-step device: first5 {[round(v) for v in d[:5]]} median {sorted(d)[25]:.1f} max {max(d):.1f}")
y = W.similar(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50): W.dwt_(y, x, wt, 13)
torch.cuda.synchronize(); print(f"dwt_ (no allocation): wall {(time.perf_counter()-t0)/50*1e6:.1f} us/step")
