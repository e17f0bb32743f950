"""Per-workgroup stamps of the last pair launch (wlbench wgtime=FILE on a -DWL_WGTIME build): distribution of start / end times.
stamps (100 MHz): 0 start, 1 prologue loads landed, 6 half of the owned steps, 2 owned steps done, 3 end (main wave 0), 4/5 level-(l+1) wave first step / end, 7 = blockIdx<<32 | xcc<<16 | hw_id"""
import sys
import numpy as np
rows = np.array([[int(v) for v in l.split()] for l in open(sys.argv[1])], dtype=np.int64)
idx, t = rows[:, 0], rows[:, 1:9]
t0 = t[:, 0].min()
us = lambda a: (a - t0) / 100.0
def q(name, a):
    a = np.sort(a)
    print(f"{name:28s} min {a[0]:7.2f}  p5 {a[len(a)//20]:7.2f}  p25 {a[len(a)//4]:7.2f}  med {a[len(a)//2]:7.2f}  p75 {a[3*len(a)//4]:7.2f}  p95 {a[19*len(a)//20]:7.2f}  max {a[-1]:7.2f}")
print(f"{len(idx)} workgroups; times in us relative to the first workgroup's start")
q("start", us(t[:, 0]))
q("prologue landed", us(t[:, 1]))
q("half of owned steps", us(t[:, 6]))
q("owned steps done", us(t[:, 2]))
q("end (main wave)", us(t[:, 3]))
q("end (level-(l+1) wave)", us(t[:, 5]))
q("prologue duration", (t[:, 1] - t[:, 0]) / 100.0)
q("first half duration", (t[:, 6] - t[:, 1]) / 100.0)
q("second half duration", (t[:, 2] - t[:, 6]) / 100.0)
q("past-chunk steps duration", (t[:, 3] - t[:, 2]) / 100.0)
q("workgroup lifetime", (t[:, 5] - t[:, 0]) / 100.0)
xcc = (t[:, 7] >> 16) & 0xf
end = us(np.maximum(t[:, 3], t[:, 5]))
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"xcc {x}: {m.sum():4d} wgs, start med {np.median(us(t[m, 0])):6.2f}, end min {end[m].min():7.2f} med {np.median(end[m]):7.2f} max {end[m].max():7.2f}")
# how many workgroups have finished by time T
for T in np.arange(np.floor(end.min()), end.max() + 1, 1.0):
    print(f"t={T:6.1f} us: {int((end <= T).sum()):5d} done")
