#!/bin/bash
O=$PWD/gpurun_out/s9; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
for n in 4096 2048; do
echo "== single level $n"
for w in 4 2 1; do for tj in 32 64 128 256; do
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=100 warm=30 check=0 opt=WL_LDS_MODE:0,WL_LDS_W:$w,WL_TJ:$tj,WL_WAVES_PER_CU:0,WL_WAVES_MIN:0
done; done
for m in 1 2 3; do for tj in 32 64 128; do
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=100 warm=30 check=0 opt=WL_LDS_MODE:$m,WL_TJ:$tj,WL_WAVES_PER_CU:0,WL_WAVES_MIN:0
done; done
done
echo "== 8192 single: W / TJ"
for w in 4 2; do for tj in 64 128 256 512; do
  timeout 60 $B L=1 mode=each reps=100 warm=30 check=0 opt=WL_LDS_MODE:0,WL_LDS_W:$w,WL_TJ:$tj,WL_WAVES_PER_CU:0,WL_WAVES_MIN:0
done; done
} > $O/bench.log 2>&1
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -4 $O/pytest.log
