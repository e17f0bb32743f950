"""time selected filters: python tools/time_filters.py batt2 batt4 batt6 [n=8192] [L=13] [dtype=f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W
from perf_matrix_util import timeit, jl
names = [a for a in sys.argv[1:] if "=" not in a]
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
n = int(kv.get("n", 8192)); L = int(kv.get("L", 13)); dt = torch.float64 if kv.get("dtype") == "f64" else torch.float32
x = jl((n, n), dt); y = W.similar(x)
for nm in names:
    for LL in sorted({1, L}):
        wt = W.wavelet(getattr(W.WT, nm))
        tf = timeit(lambda: W.dwt_oop_(y, x, wt, LL), reps=8); kf = W.last_kernel()
        ti = timeit(lambda: W.idwt_oop_(x, y, wt, LL), reps=8); ki = W.last_kernel()
        print(f"{nm} taps={len(wt.qmf)} n={n} L={LL} {kv.get('dtype','f32')} fwd {tf:.1f} us ({kf})  inv {ti:.1f} us ({ki})", flush=True)
