"""1-D and lifting transforms on non power-of-two sizes: python tools/shape_sweep2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
if os.environ.get("WL_ALT_LIB"):          # A/B runs against another build of the library
    import wavelets_jl_amd._lib as _L
    _L.LIB_PATH = os.environ["WL_ALT_LIB"]
from perf_matrix_util import timeit, jl
db4 = W.wavelet(W.WT.db4); cdf = W.wavelet(W.WT.cdf97, W.WT.Lifting)
print("| case | dtype | forward us (GB/s alg.) | inverse us (GB/s) | kernels |")
print("|---|---|---|---|---|")
for dt, tag in ((torch.float32, "f32"),):
    for n, L in ((1000000, 6), (44100 * 60, 4), (3 << 20, 20), (10000000, 7), (1 << 20, 20)):
        x = torch.randn(n, dtype=dt, device="cuda"); y = W.similar(x)
        for wt, nm in ((db4, "db4"), (cdf, "cdf97-lift")):
            tf = timeit(lambda: W.dwt_oop_(y, x, wt, L), reps=10); kf = W.last_kernel()
            ti = timeit(lambda: W.idwt_oop_(x, y, wt, L), reps=10); ki = W.last_kernel()
            b = 2 * x.numel() * x.element_size()
            print(f"| 1-D n={n} L={L} {nm} | {tag} | {tf:.1f} ({b / tf / 1e3:.0f}) | {ti:.1f} ({b / ti / 1e3:.0f}) | {kf} / {ki} |", flush=True)
    for n, L in ((1000, 3), (3000, 3), (1536, 9), (6000, 4)):
        x = jl((n, n), dt); y = W.similar(x)
        tf = timeit(lambda: W.dwt_oop_(y, x, cdf, L), reps=10); kf = W.last_kernel()
        ti = timeit(lambda: W.idwt_oop_(x, y, cdf, L), reps=10); ki = W.last_kernel()
        b = 2 * x.numel() * x.element_size()
        print(f"| 2-D lifting {n}^2 L={L} cdf97 | {tag} | {tf:.1f} ({b / tf / 1e3:.0f}) | {ti:.1f} ({b / ti / 1e3:.0f}) | {kf} / {ki} |", flush=True)
    for shape, L in (((1000, 4096), 3), ((44100, 64), 2), ((100000, 100), 5)):
        x = jl(shape, dt); y = W.similar(x)
        tf = timeit(lambda: W.dwtc_(y, x, db4, L), reps=10); kf = W.last_kernel()
        b = 2 * x.numel() * x.element_size()
        print(f"| dwtc {shape} L={L} db4 | {tag} | {tf:.1f} ({b / tf / 1e3:.0f}) | - | {kf} |", flush=True)
    for shape, L in (((100, 100, 100), 2), ((240, 240, 160), 3), ((96, 96, 96), 5)):
        x = jl(shape, dt); y = W.similar(x)
        tf = timeit(lambda: W.dwt_oop_(y, x, db4, L), reps=10); kf = W.last_kernel()
        ti = timeit(lambda: W.idwt_oop_(x, y, db4, L), reps=10); ki = W.last_kernel()
        b = 2 * x.numel() * x.element_size()
        print(f"| 3-D {shape} L={L} db4 | {tag} | {tf:.1f} ({b / tf / 1e3:.0f}) | {ti:.1f} ({b / ti / 1e3:.0f}) | {kf} / {ki} |", flush=True)
    for n, L in ((100, 2), (240, 4), (96, 5)):
        x = jl((n, n, n), dt); y = W.similar(x)
        tf = timeit(lambda: W.dwt_oop_(y, x, cdf, L), reps=10); kf = W.last_kernel()
        ti = timeit(lambda: W.idwt_oop_(x, y, cdf, L), reps=10); ki = W.last_kernel()
        b = 2 * x.numel() * x.element_size()
        print(f"| 3-D lifting {n}^3 L={L} cdf97 | {tag} | {tf:.1f} ({b / tf / 1e3:.0f}) | {ti:.1f} ({b / ti / 1e3:.0f}) | {kf} / {ki} |", flush=True)
