#!/bin/bash
O=$PWD/gpurun_out/s5; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
echo "== tail slope"
for n in 32 64 128; do for L in 1 2 3 4 5 6 7; do
  [ $((1<<L)) -le $n ] && timeout 60 $B n0=$n n1=$n L=$L mode=each reps=300 warm=50 opt=WL_NO_MULTI2D:1
done; done
echo "== 1-D tail"
for n in 4096 16384; do for L in 1 2 4 8 12; do timeout 60 $B n0=$n n1=1 L=$L mode=each reps=300 warm=50; done; done
} > $O/bench.log 2>&1
tail -3 $O/bench.log
