#!/bin/bash
# price of bit-exactness: the same library built with -ffp-contract=fast (tools/alt/, not shipped) vs the product build
O=$PWD/gpurun_out/fpc; mkdir -p $O
R=$PWD
B=$R/tools/wlbench.bin
{
for lib in wavelets.jl_amd tools/alt; do
  export LD_LIBRARY_PATH=$R/$lib:/opt/rocm/lib
  echo "=== $lib"
  for rep in 1 2; do
  timeout 60 $B n0=8192 n1=8192 L=1 mode=each reps=300 warm=100
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=300 warm=100
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=300 warm=100 filt=sym5
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=300 warm=100 fw=0
  timeout 60 $B n0=16777216 n1=1 L=24 mode=seq reps=300 warm=100
  done
done
} > $O/bench.log 2>&1
