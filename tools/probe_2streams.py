"""Scratch: independent transforms issued round-robin on two streams: the latency-bound small levels of one overlap the
bandwidth-bound first kernel of the next."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
wt = W.wavelet(W.WT.db4)
x = torch.randn(8192, 8192, dtype=torch.float32, device="cuda").t()
ys = [W.similar(x) for _ in range(4)]
def run(nstreams, steps=600):
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream()]
    for s, y in zip(streams, ys):
        with torch.cuda.stream(s):
            for _ in range(30): W.dwt_oop_(y, x, wt, 13)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        s = streams[i % nstreams]
        with torch.cuda.stream(s):
            W.dwt_oop_(ys[i % nstreams], x, wt, 13)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
for ns in (1, 2, 3, 4, 1, 2):
    print(f"{ns} stream(s): {run(ns):.4f} ms per transform", flush=True)
