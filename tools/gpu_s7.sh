#!/bin/bash
O=$PWD/gpurun_out/s7; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
echo "== tail2"
for n in 16 32 64; do
  timeout 60 $B n0=$n n1=$n L=0 mode=each reps=300 warm=50 opt=WL_NO_MULTI2D:1
  timeout 60 $B n0=$n n1=$n L=0 mode=each reps=300 warm=50 opt=WL_NO_MULTI2D:1,WL_TAIL2:0
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=300 warm=50 opt=WL_NO_MULTI2D:1
done
for t in 64 128 256 512; do timeout 60 $B n0=64 n1=64 L=0 mode=each reps=300 warm=50 opt=WL_NO_MULTI2D:1,WL_TAIL2_THREADS:$t; done
for t in 64 128 256; do timeout 60 $B n0=32 n1=32 L=0 mode=each reps=300 warm=50 opt=WL_NO_MULTI2D:1,WL_TAIL2_THREADS:$t; done
for n in 4096 1024; do timeout 60 $B n0=$n n1=1 L=0 mode=each reps=300 warm=50; timeout 60 $B n0=$n n1=1 L=0 mode=each reps=300 warm=50 opt=WL_TAIL2:0; done
echo "== chains"
for n in 128 256 512 1024 2048 4096 8192; do
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_M2D_MIN:128
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_M2D_MIN:128,WL_M2D_NL:3
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_TAIL2:0
done
timeout 60 $B n0=16777216 n1=1 L=24 reps=200 warm=50
timeout 60 $B n0=16777216 n1=1 L=24 reps=200 warm=50 opt=WL_TAIL2:0
} > $O/bench.log 2>&1
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail2 or golden or randomized or fast_and_generic" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
