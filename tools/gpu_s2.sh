#!/bin/bash
# GPU session 2: the LDS-exchange kernel -- timing of workgroup shapes, then parity.
O=$PWD/gpurun_out/s2; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
for L in 2 1 13; do
  m="mode=each"; [ $L = 13 ] && m="mode=seq"
  echo "== L=$L"
  timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS2D:0
  for mode in 0 1 2 3 4; do
    timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS_MODE:$mode
  done
  for w in 2 1; do
    timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS_MODE:0,WL_LDS_W:$w
  done
  for tj in 64 256; do
    timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS_MODE:0,WL_TJ:$tj,WL_TJ2:$tj
    timeout 60 $B L=$L $m reps=100 warm=30 opt=WL_LDS_MODE:1,WL_TJ:$tj,WL_TJ2:$tj
  done
done
echo "== other sizes / filters"
for args in "n0=4096 n1=4096 L=12" "n0=2048 n1=2048 L=11" "n0=2048 n1=2048 L=2 mode=each" "n0=1024 n1=1024 L=2 mode=each" "n0=512 n1=512 L=2 mode=each" "filt=sym5 L=13" "filt=db2 L=13" "filt=haar L=13"; do
  timeout 60 $B $args reps=100 warm=30 opt=WL_LDS2D:0
  timeout 60 $B $args reps=100 warm=30
  timeout 60 $B $args reps=100 warm=30 opt=WL_LDS_MODE:1
done
} > $O/bench.log 2>&1
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_exchange or fused_level_pair or randomized or golden or fast_and_generic" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
