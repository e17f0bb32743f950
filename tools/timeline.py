"""Launch-by-launch timeline of whole transforms from a rocprofv3 kernel trace (start / end timestamps, kernel time vs gaps, per stream).

    python tools/timeline.py trace.csv LAUNCHES_PER_TRANSFORM [NTRANSFORMS=3] [TITLE]

Prints markdown: for each of the last NTRANSFORMS transforms in the trace one row per launch (start and end relative to the first
launch's start, duration, gap to the previous launch's end, queue / stream id, grid, kernel), then the sums (kernel time, gaps, wall).
"""
import csv
import sys


def main():
    path, per = sys.argv[1], int(sys.argv[2])
    ntr = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    title = sys.argv[4] if len(sys.argv) > 4 else path
    rows = [r for r in csv.DictReader(open(path)) if "wl::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-per * ntr:]
    print(f"### {title}\n")
    walls, ktimes = [], []
    for t in range(ntr):
        grp = rows[t * per:(t + 1) * per]
        if not grp:
            continue
        t0 = int(grp[0]["Start_Timestamp"])
        prev_end = int(rows[t * per - 1]["End_Timestamp"]) if t > 0 else None
        print(f"transform {t + 1} of the last {ntr}\n")
        print("| # | start µs | end µs | kernel µs | gap before µs | queue | grid x wg | kernel |")
        print("|---|---|---|---|---|---|---|---|")
        ksum = 0.0
        last_end = prev_end
        for i, r in enumerate(grp):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = (s - last_end) / 1e3 if last_end is not None else float("nan")
            last_end = max(e, last_end) if last_end is not None else e
            ksum += (e - s) / 1e3
            name = r["Kernel_Name"].replace("void wl::", "").split("(")[0][:48]
            q = r.get("Queue_Id", "?")
            print(f"| {i + 1} | {(s - t0) / 1e3:.1f} | {(e - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} | {q} | "
                  f"{r['Grid_Size_X']}x{r['Grid_Size_Y']} / {r['Workgroup_Size_X']} | `{name}` |")
        wall = (max(int(r["End_Timestamp"]) for r in grp) - t0) / 1e3
        walls.append(wall)
        ktimes.append(ksum)
        print(f"\nkernel time {ksum:.1f} µs, first start → last end {wall:.1f} µs, gaps inside {wall - ksum:.1f} µs\n")
    if len(rows) >= 2 * per:
        period = (int(rows[-per]["Start_Timestamp"]) - int(rows[-per * ntr]["Start_Timestamp"])) / 1e3 / (ntr - 1) if ntr > 1 else float("nan")
        print(f"period between transforms (start to start, mean of the last {ntr}): {period:.1f} µs\n")


if __name__ == "__main__":
    main()
