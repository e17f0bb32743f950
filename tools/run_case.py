"""Tiny driver for profiling one entry point under rocprofv3: python tools/run_case.py {idwt2d|lift2d|dwt3d|modwt} [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavelets_jl_amd as W

case = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for kv in sys.argv[3:]:                       # per-context options: KEY:VALUE
    k, v = kv.split(":")
    W.set_option(k, int(v))
g = torch.Generator(device="cpu").manual_seed(7)
db4 = W.wavelet(W.WT.db4)
if case == "idwt2d":
    x = torch.randn(8192, 8192, generator=g, dtype=torch.float32).cuda().t()
    y = W.similar(x)
    fn = lambda: W.idwt_oop_(y, x, db4, 13)
elif case == "lift2d":
    x = torch.randn(8192, 8192, generator=g, dtype=torch.float32).cuda().t()
    y = W.similar(x)
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    fn = lambda: W.dwt_oop_(y, x, sch, 13)
elif case == "lift2d_inv":
    x = torch.randn(8192, 8192, generator=g, dtype=torch.float32).cuda().t()
    y = W.similar(x)
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    fn = lambda: W.idwt_oop_(y, x, sch, 13)
elif case == "dwt2d_f64":
    x = torch.randn(8192, 8192, generator=g, dtype=torch.float64).cuda().t()
    y = W.similar(x)
    fn = lambda: W.dwt_oop_(y, x, db4, 13)
elif case == "dwt2d_db8":
    x = torch.randn(8192, 8192, generator=g, dtype=torch.float32).cuda().t()
    y = W.similar(x)
    db8 = W.wavelet(W.WT.db8)
    fn = lambda: W.dwt_oop_(y, x, db8, 13)
elif case in ("idwt2d_sym8", "idwt2d_sym5"):      # k_inv2d_lds_long: 16 taps / 10 taps (the default wavelet of denoise)
    x = torch.randn(8192, 8192, generator=g, dtype=torch.float32).cuda().t()
    y = W.similar(x)
    wt = W.wavelet(W.WT.sym8 if case.endswith("sym8") else W.WT.sym5)
    fn = lambda: W.idwt_oop_(y, x, wt, 13)
elif case == "lift3d":
    x = torch.randn(256, 256, 256, generator=g, dtype=torch.float32).cuda().permute(2, 1, 0)
    y = W.similar(x)
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    fn = lambda: W.dwt_oop_(y, x, sch, 8)
elif case == "idwt2d_f64":
    x = torch.randn(8192, 8192, generator=g, dtype=torch.float64).cuda().t()
    y = W.similar(x)
    fn = lambda: W.idwt_oop_(y, x, db4, 13)
elif case == "lift1d_inv":
    x = torch.randn(1 << 24, generator=g, dtype=torch.float32).cuda()
    y = W.similar(x)
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    fn = lambda: W.idwt_oop_(y, x, sch, 24)
elif case == "img1080":
    x = torch.randn(1920, 1080, generator=g, dtype=torch.float32).cuda().t()
    y = W.similar(x)
    fn = lambda: W.dwt_oop_(y, x, db4, 3)
elif case == "img1080_inv":
    x = torch.randn(1920, 1080, generator=g, dtype=torch.float32).cuda().t()
    y = W.similar(x)
    fn = lambda: W.idwt_oop_(y, x, db4, 3)
elif case == "lift1d_1e6":
    x = torch.randn(1000000, generator=g, dtype=torch.float32).cuda()
    y = W.similar(x)
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    fn = lambda: W.dwt_oop_(y, x, sch, 6)
elif case == "dwt3d":
    x = torch.randn(512, 512, 512, generator=g, dtype=torch.float32).cuda().permute(2, 1, 0)
    y = W.similar(x)
    fn = lambda: W.dwt_oop_(y, x, db4, 9)
elif case == "idwt3d":
    x = torch.randn(512, 512, 512, generator=g, dtype=torch.float32).cuda().permute(2, 1, 0)
    y = W.similar(x)
    fn = lambda: W.idwt_oop_(y, x, db4, 9)
elif case == "batt6":                        # 59 taps: the two-pass line kernels of wl_vlong.hip (k_vl_lines)
    x = torch.randn(8192, 8192, generator=g, dtype=torch.float32).cuda().t()
    y = W.similar(x)
    wt = W.wavelet(W.WT.batt6)
    fn = lambda: W.dwt_oop_(y, x, wt, 13)
elif case == "modwt":
    x = torch.randn(1 << 24, generator=g, dtype=torch.float32).cuda()
    fn = lambda: W.modwt(x, db4, 8)
elif case == "c2":
    x = torch.randn(1 << 24, generator=g, dtype=torch.float32).cuda()
    y = W.similar(x)
    fn = lambda: W.dwt_oop_(y, x, db4, 24)
elif case == "c4":
    x = torch.randn(1 << 24, generator=g, dtype=torch.float32).cuda()
    y = W.similar(x)
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    fn = lambda: W.dwt_oop_(y, x, sch, 24)
elif case == "c5":
    x = torch.randn(8192, 1 << 16, generator=g, dtype=torch.float32).cuda().t()
    y = W.similar(x)
    fn = lambda: W.dwtc_(y, x, db4, 16)
elif case == "c1":
    x = torch.rand(1 << 20, generator=g, dtype=torch.float64).cuda()
    y = W.similar(x)
    fn = lambda: W.dwt_oop_(y, x, W.wavelet(W.WT.db2), 20)
elif case == "lift1d_l3":                    # one launch of the C4 dominant kernel (k_lift1d_fwd3: levels 1-3 fused)
    x = torch.randn(1 << 24, generator=g, dtype=torch.float32).cuda()
    y = W.similar(x)
    sch = W.wavelet(W.WT.cdf97, W.WT.Lifting)
    fn = lambda: W.dwt_oop_(y, x, sch, 3)
elif case == "wpt":
    x = torch.randn(1 << 22, generator=g, dtype=torch.float32).cuda()
    y = W.similar(x)
    fn = lambda: W.wpt_(y, x, db4, 6)
elif case == "batch2d":
    x = torch.randn(64, 2048, 2048, generator=g, dtype=torch.float32).cuda().permute(2, 1, 0)
    y = W.similar(x)
    fn = lambda: W.dwt_batch(x, db4, 11, y=y)
elif case == "denoise":
    x = torch.randn(2048, 2048, generator=g, dtype=torch.float32).cuda().t()
    fn = lambda: W.denoise(x, TI=True)
else:
    raise SystemExit("unknown case")
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print(case, "done", W.last_kernel())
