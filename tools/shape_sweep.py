"""Throughput of the 2-D filter transform on image-like (non power-of-two) shapes: python tools/shape_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
from perf_matrix_util import timeit, jl
shapes = [((1080, 1920), 3), ((2160, 3840), 4), ((3000, 4000), 3), ((4000, 6000), 4), ((1000, 1000), 3), ((5000, 5000), 3), ((6000, 6000), 4),
          ((4096, 4096), 12), ((1024, 768), 8), ((7680, 4320), 5)]
print("| shape | L | filter | dtype | forward us (GB/s alg.) | inverse us (GB/s) | kernels |")
print("|---|---|---|---|---|---|---|")
for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
    for shape, L in shapes:
        x = jl(shape, dt); y = W.similar(x)
        for nm in ("db4", "sym5"):
            wt = W.wavelet(getattr(W.WT, nm))
            tf = timeit(lambda: W.dwt_oop_(y, x, wt, L), reps=10); kf = W.last_kernel()
            ti = timeit(lambda: W.idwt_oop_(x, y, wt, L), reps=10); ki = W.last_kernel()
            b = 2 * x.numel() * x.element_size()
            print(f"| {shape[0]}x{shape[1]} | {L} | {nm} | {tag} | {tf:.1f} ({b / tf / 1e3:.0f}) | {ti:.1f} ({b / ti / 1e3:.0f}) | {kf} / {ki} |", flush=True)
        del x, y
