#!/bin/bash
# Timeline of the headline transform (8192^2 db4 f32, L = 13, three rotating inputs): every launch's start / end from a rocprofv3
# kernel trace.  Run on the GPU box from the repo root: bash tools/r06_timeline.sh [outdir-name] [extra wlbench args]
R=$PWD; O=$R/gpurun_out/${1:-timeline}; mkdir -p $O
B=$R/tools/wlbench.bin
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
$R/tools/rp.sh $O/trace r06 "--kernel-trace" $B n0=8192 n1=8192 L=13 rot=3 reps=60 warm=20 check=0 ${@:2}
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
n=$(python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wl::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# launches per transform = distance between the last two launches of the biggest grid
big = max(int(r["Grid_Size_X"]) for r in rows)
idx = [i for i, r in enumerate(rows) if int(r["Grid_Size_X"]) == big]
print(idx[-1] - idx[-2])
PY
)
python3 $R/tools/timeline.py "$f" $n 3 "8192^2 db4 f32 L=13, wlbench rot=3 back to back, rocprofv3 --kernel-trace" > $O/timeline.md
cat $O/timeline.md
find $O -name "*.csv" -size +4M -delete
