"""Scratch: the whole C5 batch (65536 signals x 2^16 samples = 2^32 Float32 elements, 16 GiB) on ONE GPU: 64-bit
indexing beyond 2^32 elements, slab launches, round trip, and per-column agreement with the 1-D transform."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
wt = W.wavelet(W.WT.db4)
n, ns = 1 << 16, 65536
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.empty(ns, n, dtype=torch.float32, device="cuda").normal_(generator=g).t()      # Julia layout (n, ns)
print("elements", x.numel(), "GiB", x.numel() * 4 / 2**30, flush=True)
torch.cuda.synchronize(); t = time.perf_counter()
y = W.dwtc(x, wt, 16)
torch.cuda.synchronize(); print(f"dwtc: {(time.perf_counter()-t)*1e3:.1f} ms (first call, incl. workspace allocation)  kernel={W.last_kernel()}", flush=True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); W.dwtc_(y, x, wt, 16); ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1)
print(f"dwtc (steady): {ms:.2f} ms = {2*x.numel()*4/ms/1e6:.0f} GB/s algorithmic", flush=True)
for j in (0, 1, 32767, 32768, 65534, 65535):                                            # columns on both sides of the slab boundary
    yj = W.dwt(x[:, j].contiguous(), wt, 16)
    assert torch.equal(yj, y[:, j]), j
print("columns agree with the 1-D transform bit for bit", flush=True)
xr = W.idwtc(y, wt, 16)
err = (xr - x).abs().max().item()
print("round trip max abs err", err)
assert err < 1e-4
print("OK")
