#!/bin/bash
O=$PWD/gpurun_out/s15; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
{
for nl in 4 5 6; do for ts in 2048 4096 8192; do
  timeout 60 $B n0=16777216 n1=1 L=24 mode=seq reps=300 warm=100 opt=WL_NLMAX:$nl,WL_TS:$ts
done; done
for nl in 4 6; do
  timeout 60 $B n0=1048576 n1=1 L=20 mode=seq reps=300 warm=100 dtype=f64 filt=db2 opt=WL_NLMAX:$nl
  timeout 60 $B n0=65536 n1=8192 nd=1 L=16 mode=seq reps=50 warm=20 opt=WL_NLMAX:$nl
done
} > $O/bench.log 2>&1
