"""One transform's chain from per-workgroup stamps (wlbench wgtime dumps of pair / tileB / tile / tail of the same run): python tools/wgtime_chain.py DIR"""
import sys, os
import numpy as np
d = sys.argv[1]
def load(n):
    p = os.path.join(d, n + ".txt")
    rows = np.array([[int(v) for v in l.split()] for l in open(p)], dtype=np.int64)
    return rows[:, 1:9]
pair, tb, t3, tl = load("pair"), load("tileB"), load("tile"), load("tail")
t0 = pair[:, 0].min()
us = lambda a: (a - t0) / 100.0
def q(name, a):
    a = np.sort(np.asarray(a, dtype=float))
    n = len(a)
    print(f"{name:44s} min {a[0]:7.2f} p25 {a[n//4]:7.2f} med {a[n//2]:7.2f} p75 {a[3*n//4]:7.2f} max {a[-1]:7.2f}")
print("times in us from the first pair workgroup's start")
q("pair: wg start", us(pair[:, 0])); q("pair: wg end", us(np.maximum(pair[:, 3], pair[:, 5])))
q("tileB: wg start", us(tb[:, 0])); q("tileB: loads + dim-2 of level a issued", us(tb[:, 1])); q("tileB: after barrier 1", us(tb[:, 2]))
q("tileB: level a dim-1 done (barrier 2)", us(tb[:, 3])); q("tileB: end", us(tb[:, 4]))
q("tileB: dur load+dim2", (tb[:, 1] - tb[:, 0]) / 100.); q("tileB: dur barrier wait", (tb[:, 2] - tb[:, 1]) / 100.)
q("tileB: dur dim1 a", (tb[:, 3] - tb[:, 2]) / 100.); q("tileB: dur level b", (tb[:, 4] - tb[:, 3]) / 100.); q("tileB: lifetime", (tb[:, 4] - tb[:, 0]) / 100.)
q("tile3: wg start", us(t3[:, 0])); q("tile3: staged", us(t3[:, 1])); q("tile3: level a done", us(t3[:, 2])); q("tile3: level b done", us(t3[:, 3])); q("tile3: end", us(t3[:, 4]))
q("tile3: dur stage", (t3[:, 1] - t3[:, 0]) / 100.); q("tile3: dur a", (t3[:, 2] - t3[:, 1]) / 100.); q("tile3: dur b", (t3[:, 3] - t3[:, 2]) / 100.); q("tile3: dur c", (t3[:, 4] - t3[:, 3]) / 100.)
r = tl[0]
print("tail2: start %.2f staged %.2f " % (us(r[0]), us(r[1])) + " ".join("lev%d %.2f" % (k, us(r[1 + k])) for k in range(1, 6)) + " end %.2f" % us(r[7]))
print("gaps: pair last end -> tileB first start %.2f, tileB last end -> tile3 first start %.2f, tile3 last end -> tail start %.2f" % (
    us(tb[:, 0].min()) - us(np.maximum(pair[:, 3], pair[:, 5]).max()), us(t3[:, 0].min()) - us(tb[:, 4].max()), us(r[0]) - us(t3[:, 4].max())))
