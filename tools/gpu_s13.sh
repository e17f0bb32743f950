#!/bin/bash
O=$PWD/gpurun_out/s13; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
for t in 256 512 1024; do
for n in 256 1024 2048; do for L in 1 2 3; do timeout 60 $B n0=$n n1=$n L=$L mode=each reps=200 warm=50 opt=WL_TILE_THREADS:$t,WL_TILE_NL3_MAX:4096,WL_LDS2D_MIN_ROWS:1000000; done; done
done
for t in 512 1024; do
for n in 128 256 512 1024 2048 4096 8192; do
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_TILE_THREADS:$t,WL_TILE_NL3_MAX:0
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_TILE_THREADS:$t,WL_TILE_NL3_MAX:1024
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_TILE_THREADS:$t,WL_TILE_NL3_MAX:0,WL_TILE_MAX:1024
done; done
} > $O/bench.log 2>&1
