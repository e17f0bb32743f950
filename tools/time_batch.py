"""Batch of images: one wl_dwt_filter_batch chain against B single transforms (GPU box).  Markdown rows."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavelets_jl_amd as W


def t_us(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3


db4 = W.wavelet(W.WT.db4)
print("| images | L | batch us | kernel | B single calls us | speed-up | batch GB/s (algorithmic) | idwt batch us |")
print("|---|---|---|---|---|---|---|---|")
for n, nb, L in ((2048, 64, 4), (2048, 64, 11), (1024, 64, 4), (1024, 256, 10), (512, 256, 4), (512, 1024, 9), (256, 1024, 8)):
    xb = torch.randn(nb, n, n, dtype=torch.float32, device="cuda").permute(2, 1, 0)
    yb = W.similar(xb)
    zb = W.similar(xb)
    fb = lambda: W.dwt_batch(xb, db4, L, y=yb)
    tb = t_us(fb)
    kb = W.last_kernel()
    ti = t_us(lambda: W.idwt_batch(yb, db4, L, y=zb))
    ys = W.similar(xb[:, :, 0])
    def fs():
        for i in range(nb):
            W.dwt_oop_(ys, xb[:, :, i], db4, L)
    ts = t_us(fs, reps=5)
    gb = 2 * xb.numel() * 4 / tb / 1e3
    print(f"| {nb} x {n}^2 | {L} | {tb:.1f} | {kb} | {ts:.1f} | {ts / tb:.2f} | {gb:.0f} | {ti:.1f} |")
