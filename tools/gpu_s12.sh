#!/bin/bash
O=$PWD/gpurun_out/s12; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
for n in 128 256 512 1024 2048 4096 8192; do
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_TILE:0
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_TILE_NL3_MAX:0
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_TILE_NL3_MAX:4096
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_TILE_MAX:1024
  timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=WL_TILE_MAX:4096
done
for n in 256 512 1024 2048; do for L in 1 2 3; do timeout 60 $B n0=$n n1=$n L=$L mode=each reps=200 warm=50 opt=WL_TILE_NL3_MAX:4096,WL_LDS2D_MIN_ROWS:1000000; done; done
} > $O/bench.log 2>&1
$R/tools/rp.sh $O/stats_L13 st "--kernel-trace --stats" $B L=13 reps=200 warm=50 check=0
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile or golden or fast_and_generic or randomized or lds_exchange or fused_level_pair or full_size" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
