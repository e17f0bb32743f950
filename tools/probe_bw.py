"""Scratch measurement (not product code): level-1 launch time of the 2-D kernel vs a torch copy and
vs the no-arithmetic probe variant (WL_PROBE=1) on the 8192^2 f32 headline array."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W

def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in ev)
    return d[len(d)//2] * 1e3, d[0] * 1e3

n = int(os.environ.get("N", "8192"))
x = torch.randn(n, n, dtype=torch.float32, device="cuda").t()
y = W.similar(x)
wt = W.wavelet(W.WT.db4)
W.reserve_workspace(x, 1)
nbytes = 2 * x.numel() * 4
out = {}
med, mn = timeit(lambda: y.copy_(x)); out["torch_copy"] = (med, mn, nbytes / med / 1e3)
med, mn = timeit(lambda: W.dwt_(y, x, wt, 1)); out["dwt_L1_%s" % os.environ.get("WL_PROBE", "0")] = (med, mn, nbytes / med / 1e3)
print(json.dumps({k: [round(v, 1) for v in t] for k, t in out.items()}), "tj", os.environ.get("WL_TJ"))
