#!/bin/bash
O=$PWD/gpurun_out/s10; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
timeout 60 $B L=1 mode=each reps=100 warm=30
timeout 60 $B L=1 mode=each reps=100 warm=30 opt=WL_TJ:256
timeout 60 $B L=1 mode=each reps=100 warm=30 opt=WL_TJ:64
timeout 60 $B L=1 mode=each reps=100 warm=30 opt=WL_LDS_MODE:1
for o in "WL_LDS_PAIR_MIN:0,WL_LDS_W:2" "WL_LDS_PAIR_MIN:0,WL_LDS_MODE:1" "WL_LDS_PAIR_MIN:0,WL_LDS_W:4" "WL_LDS_PAIR_MIN:0,WL_LDS_W:2,WL_TJ2:256" "WL_LDS_PAIR_MIN:0,WL_LDS_MODE:1,WL_TJ2:256" "WL_LDS_PAIR_MIN:0,WL_LDS_W:2,WL_TJ2:64"; do
  timeout 60 $B L=2 mode=each reps=100 warm=30 opt=$o
done
for n in 4096 2048 1024; do
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=100 warm=30 opt=WL_M2D_MAX:128
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=100 warm=30 opt=WL_M2D_MAX:128,WL_LDS_W:2
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=100 warm=30 opt=WL_M2D_MAX:128,WL_LDS_MODE:1
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=100 warm=30 opt=WL_M2D_MAX:128,WL_WAVES_PER_CU:4
  timeout 60 $B n0=$n n1=$n L=2 mode=each reps=100 warm=30 opt=WL_M2D_MAX:128,WL_LDS_PAIR_MIN:0,WL_LDS_W:2
done
timeout 60 $B L=13 reps=200 warm=50
timeout 60 $B L=13 reps=200 warm=50 opt=WL_LDS_PAIR_MIN:0,WL_LDS_W:2
timeout 60 $B n0=4096 n1=4096 L=12 reps=200 warm=50
timeout 60 $B n0=2048 n1=2048 L=11 reps=200 warm=50
} > $O/bench.log 2>&1
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_exchange or full_size or golden or fast_and_generic or randomized or cube" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
