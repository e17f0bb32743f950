#!/bin/bash
O=$PWD/gpurun_out/s6; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
for cfg in "1 WL_LDS_MODE:0" "2 WL_LDS_MODE:0,WL_LDS_W:2" "2 WL_LDS_MODE:1"; do
  set -- $cfg
  echo "== L=$1 $2"
  for dbg in 0 1 2 4 8 16 12 3 28 29 31; do
    timeout 60 $B L=$1 mode=each reps=60 warm=20 check=0 opt=$2,WL_LDS_DBG:$dbg
  done
done
} > $O/bench.log 2>&1
