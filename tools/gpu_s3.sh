#!/bin/bash
O=$PWD/gpurun_out/s3; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
for L in 2 1 13; do
  m="mode=each"; [ $L = 13 ] && m="mode=seq"
  echo "== L=$L"
  timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS2D:0
  for mode in 0 1 2 3; do
    timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS_MODE:$mode
  done
  timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS_MODE:0,WL_LDS_W:2
  timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS_MODE:1,WL_TJ:64,WL_TJ2:64
  timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS_MODE:0,WL_TJ:256,WL_TJ2:256
  timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_FUSE2:0
done
} > $O/bench.log 2>&1
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_exchange" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
