#!/bin/bash
# calibration: current C3 by depth, per-launch trace
R=$PWD; O=$R/gpurun_out/e0; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
{
for L in 13 1 2 3 4 5 13; do timeout 60 $B L=$L mode=seq reps=300 warm=300 check=0; done
timeout 60 $B L=13 mode=each reps=200 warm=300 check=0
timeout 60 $B L=13 mode=seq reps=300 warm=300 check=0 fw=0
} > $O/bench.log 2>&1
$R/tools/rp.sh $O/trace e0 "--kernel-trace" $B L=13 reps=40 warm=20 check=0
python3 $R/tools/trace_levels.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 12 > $O/levels.txt 2>&1
find $O -name "*.csv" -size +2M -delete
cat $O/bench.log $O/levels.txt
