#!/bin/bash
O=$PWD/gpurun_out/s8; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
echo "== pair kernel"
for o in "WL_LDS_PAIR_MIN:0,WL_LDS_MODE:0,WL_LDS_W:2" "WL_LDS_PAIR_MIN:0,WL_LDS_MODE:1" "WL_LDS_PAIR_MIN:0,WL_LDS_MODE:0,WL_LDS_W:4" "WL_LDS_PAIR_MIN:0,WL_LDS_MODE:0,WL_LDS_W:2,WL_TJ2:256" "WL_LDS_PAIR_MIN:0,WL_LDS_MODE:1,WL_TJ2:256" "WL_LDS_PAIR_MIN:0,WL_LDS_MODE:2" "WL_LDS_MODE:0"; do
  timeout 60 $B L=2 mode=each reps=100 warm=30 opt=$o
done
timeout 60 $B L=1 mode=each reps=100 warm=30
timeout 60 $B L=13 reps=200 warm=50
timeout 60 $B L=13 reps=200 warm=50 opt=WL_LDS_PAIR_MIN:0,WL_LDS_MODE:0,WL_LDS_W:2
timeout 60 $B L=13 reps=200 warm=50 opt=WL_LDS_PAIR_MIN:67108864,WL_LDS_MODE:0,WL_LDS_W:2
timeout 60 $B L=13 reps=200 warm=50 opt=WL_M2D_MIN:128
} > $O/bench.log 2>&1
$R/tools/rp.sh $O/stats_L13 st "--kernel-trace --stats" $B L=13 reps=200 warm=50 check=0
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -4 $O/pytest.log
