"""Device timings of the packet transforms (wpt / iwpt): python tools/time_wpt.py  (GPU box).  Markdown rows."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavelets_jl_amd as W


def t_us(fn, reps=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3


g = torch.Generator(device="cpu").manual_seed(3)
db4 = W.wavelet(W.WT.db4)
cdf = W.wavelet(W.WT.cdf97, W.WT.Lifting)
print("| case | us (fast) | kernel | us (per-depth tier) |")
print("|---|---|---|---|")
for pw, L in ((22, 6), (22, 22), (18, 18), (16, 16), (14, 14), (24, 4)):
    n = 1 << pw
    x = torch.randn(n, generator=g, dtype=torch.float32).cuda()
    y = W.similar(x)
    tree = L                      # wpt!(y, x, wt, L::Integer): the full tree by depth (wl_wpt_*_full)
    W.reserve_workspace(x, L, full=True)
    for name, fn in (("wpt db4", lambda: W.wpt_(y, x, db4, tree)), ("iwpt db4", lambda: W.iwpt_(y, x, db4, tree)),
                     ("wpt cdf9/7", lambda: W.wpt_(y, cdf, tree)), ("iwpt cdf9/7", lambda: W.iwpt_(y, cdf, tree))):
        fast = t_us(fn)
        k = W.last_kernel()
        W.set_option("WL_WPT_FAST", 0)
        slow = t_us(fn)
        W.clear_options()
        print(f"| {name} 2^{pw} depth {L} f32 | {fast:.1f} | {k} | {slow:.1f} |")

# ---- partially split trees (round 5: the packet kernels take the node bits as a per-segment split mask) ----
import numpy as np
print()
print("| tree (2^22, db4, f32) | wpt us | kernel | iwpt us | kernel | wpt us, per-depth tier |")
print("|---|---|---|---|---|---|")
n = 1 << 22
x = torch.randn(n, generator=g, dtype=torch.float32).cuda()
y = W.similar(x)
rs = np.random.default_rng(5)


def rand_tree(depth, p):
    t = np.zeros(n - 1, dtype=np.uint8)
    t[0] = 1
    for i in range(1, 2 ** depth - 1):
        t[i] = 1 if (t[(i + 1) // 2 - 1] and rs.random() < p) else 0
    return t


for label, tree in (("full, depth 9", W.maketree(n, 9, "full")), ("dwt-shaped, depth 9", W.maketree(n, 9, "dwt")), ("random (p = 0.6), depth 9", rand_tree(9, 0.6)),
                    ("full, depth 22", W.maketree(n, 22, "full")), ("dwt-shaped, depth 22", W.maketree(n, 22, "dwt")), ("random (p = 0.8), depth 16", rand_tree(16, 0.8))):
    W.reserve_workspace(x, 22, full=True)
    tf = t_us(lambda: W.wpt_(y, x, db4, tree)); kf = W.last_kernel()
    ti = t_us(lambda: W.iwpt_(y, x, db4, tree)); ki = W.last_kernel()
    W.set_option("WL_WPT_FAST", 0)
    ts = t_us(lambda: W.wpt_(y, x, db4, tree))
    W.clear_options()
    print(f"| {label} | {tf:.1f} | {kf} | {ti:.1f} | {ki} | {ts:.1f} |")
