"""per-launch durations of the last transform in a rocprofv3 kernel trace: python tools/trace_levels.py trace.csv [ncalls_per_transform]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "wl::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t0 = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - t0) / 1e3 if t0 else 0.0
    t0 = e
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:<6} wg {r['Workgroup_Size_X']:>4} vgpr {r['VGPR_Count']:>3} lds {r['LDS_Block_Size']:>6}  {r['Kernel_Name'][:70]}")
