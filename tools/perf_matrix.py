"""Device timings of every public entry point at representative sizes (median of HIP-event pairs around single calls,
output/workspace pre-allocated).  Writes a markdown table (stdout).  Run on the GPU box:
    python tools/perf_matrix.py > gpurun_out/perf_matrix.md"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavelets_jl_amd as W

if "--arithmetic" in sys.argv:          # "fused": the opt-in FMA build (tools/fp_contract_price.sh prices it against "exact")
    W.set_arithmetic(sys.argv[sys.argv.index("--arithmetic") + 1])


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in ev)
    return d[len(d) // 2] * 1e3


def jl(shape, dtype):
    t = torch.randn(*reversed(shape), dtype=dtype, device="cuda")
    return t.permute(*reversed(range(len(shape)))) if len(shape) > 1 else t


rows = []
db4, sym5, db8 = W.wavelet(W.WT.db4), W.wavelet(W.WT.sym5), W.wavelet(W.WT.db8)
cdf = W.wavelet(W.WT.cdf97, W.WT.Lifting)
for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
    es = 4 if dtype == torch.float32 else 8
    for label, shape, wt, L in (("1-D 2^24", (1 << 24,), db4, 24), ("1-D 2^20", (1 << 20,), db4, 20), ("2-D 8192^2", (8192, 8192), db4, 13),
                                ("2-D 2048^2", (2048, 2048), db4, 11), ("3-D 256^3", (256, 256, 256), db4, 8), ("3-D 512^3", (512, 512, 512), db4, 9),
                                ("2-D 8192^2 sym5", (8192, 8192), sym5, 13), ("2-D 8192^2 db8 (16 taps)", (8192, 8192), db8, 13),
                                ("1-D 2^24 cdf9/7 lifting", (1 << 24,), cdf, 24), ("2-D 4096^2 cdf9/7 lifting", (4096, 4096), cdf, 12), ("2-D 8192^2 cdf9/7 lifting", (8192, 8192), cdf, 13),
                                ("3-D 256^3 cdf9/7 lifting", (256, 256, 256), cdf, 8)):
        if dtype == torch.float64 and shape == (8192, 8192) and wt is sym5:
            continue
        x = jl(shape, dtype); y = W.similar(x)
        tf = timeit(lambda: W.dwt_oop_(y, x, wt, L)); kf = W.last_kernel()
        ti = timeit(lambda: W.idwt_oop_(x, y, wt, L)); ki = W.last_kernel()
        alg = 2 * x.numel() * es
        rows.append((f"dwt / idwt {label}", tag, L, tf, ti, alg / tf / 1e3, alg / ti / 1e3, f"{kf} / {ki}"))
        del x, y
    x = jl((1 << 16, 2048), dtype); y = W.similar(x)
    tf = timeit(lambda: W.dwtc_(y, x, db4, 16)); kf = W.last_kernel()
    ti = timeit(lambda: W.idwtc_(x, y, db4, 16)); ki = W.last_kernel()
    alg = 2 * x.numel() * es
    rows.append(("dwtc / idwtc 2048 signals x 2^16", tag, 16, tf, ti, alg / tf / 1e3, alg / ti / 1e3, f"{kf} / {ki}"))
    del x, y
    v = jl((1 << 22,), dtype)
    tree = W.maketree(1 << 22, 6, "full")
    out = W.similar(v)
    tf = timeit(lambda: W.wpt_(out, v, db4, tree)); kf = W.last_kernel()
    ti = timeit(lambda: W.iwpt_(v, out, db4, tree)); ki = W.last_kernel()
    alg = 2 * v.numel() * es * 6
    rows.append(("wpt / iwpt 2^22, full tree depth 6", tag, 6, tf, ti, alg / tf / 1e3, alg / ti / 1e3, f"{kf} / {ki} (traffic = 6 levels)"))
    v = jl((1 << 22,), dtype)
    w = W.modwt(v, db4, 8)
    tf = timeit(lambda: W.modwt(v, db4, 8)); ti = timeit(lambda: W.imodwt(w, db4))
    alg = 3 * v.numel() * es * 8
    rows.append(("modwt / imodwt 2^22 x 8 levels", tag, 8, tf, ti, alg / tf / 1e3, alg / ti / 1e3, "k_modwt_step / k_imodwt_step (traffic = 3N per level)"))
    t1 = timeit(lambda: W.denoise(v), reps=5)
    a = jl((2048, 2048), dtype)
    t2 = timeit(lambda: W.denoise(a), reps=5)
    rows.append(("denoise 1-D 2^22 / 2-D 2048^2 (sym5, VisuShrink, L=6)", tag, 6, t1, t2, float("nan"), float("nan"), "noisest + dwt + threshold! + idwt"))
    del v, w, a, out
    torch.cuda.empty_cache()
# ---- every orthogonal filter of the reference's FILTERS table (wt_main.jl:372-436) on the headline shape ----
frows = []
names = ["haar"] + [f"db{i}" for i in range(2, 11)] + ["coif2", "coif4", "coif6", "coif8"] + [f"sym{i}" for i in range(4, 11)] + \
        ["batt2", "batt4", "batt6", "beyl", "vaid"]
x = jl((8192, 8192), torch.float32); y = W.similar(x)
for nm in names:
    wt = W.wavelet(getattr(W.WT, nm))
    tf = timeit(lambda: W.dwt_oop_(y, x, wt, 13), reps=8); kf = W.last_kernel()
    ti = timeit(lambda: W.idwt_oop_(x, y, wt, 13), reps=8); ki = W.last_kernel()
    frows.append((nm, len(wt.qmf), tf, ti, kf, ki))
del x, y
torch.cuda.empty_cache()

print("| entry point | T | L | forward µs | inverse µs | fwd GB/s (algorithmic) | inv GB/s | dominant kernels |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.0f} | {r[6]:.0f} | {r[7]} |")

print()
print("Filter table (wt_main.jl:372-436), 2-D dwt / idwt 8192 x 8192 Float32, L = 13:")
print()
print("| filter | taps | forward µs | inverse µs | forward kernel | inverse kernel |")
print("|---|---|---|---|---|---|")
for r in frows:
    print(f"| {r[0]} | {r[1]} | {r[2]:.1f} | {r[3]:.1f} | {r[4]} | {r[5]} |")
