#!/bin/bash
O=$PWD/gpurun_out/s4; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
echo "== small-level chains"
for n in 4096 2048 1024 512 256 128; do
  for o in "WL_LDS2D:0" "WL_LDS_MODE:1" "WL_LDS_MODE:0,WL_LDS_W:2" \
           "WL_LDS_MODE:1,WL_M2D_MAX:512" "WL_LDS_MODE:1,WL_M2D_MAX:1024" "WL_LDS_MODE:1,WL_M2D_MAX:512,WL_M2D_NL:3" "WL_LDS_MODE:1,WL_M2D_MAX:1024,WL_M2D_NL:3" \
           "WL_LDS_MODE:1,WL_M2D_MAX:512,WL_M2D_MIN:256" "WL_LDS_MODE:1,WL_M2D_MAX:1024,WL_M2D_MIN:256" "WL_LDS_MODE:1,WL_M2D_MAX:1024,WL_M2D_MIN:256,WL_M2D_NL:3" \
           "WL_LDS_MODE:1,WL_LDS_PAIR_MIN:0" "WL_LDS_MODE:1,WL_LDS_PAIR_MIN:0,WL_M2D_MIN:256"; do
    timeout 60 $B n0=$n n1=$n L=0 reps=200 warm=50 opt=$o
  done
done
echo "== single launches"
for n in 4096 2048 1024 512 256; do
  timeout 60 $B n0=$n n1=$n L=2 mode=each reps=200 warm=50 opt=WL_LDS_MODE:1,WL_LDS_PAIR_MIN:0
  timeout 60 $B n0=$n n1=$n L=2 mode=each reps=200 warm=50 opt=WL_LDS_MODE:0,WL_LDS_W:2,WL_LDS_PAIR_MIN:0
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=200 warm=50 opt=WL_LDS_MODE:1
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=200 warm=50 opt=WL_LDS_MODE:0
  timeout 60 $B n0=$n n1=$n L=1 mode=each reps=200 warm=50 opt=WL_LDS2D:0
  timeout 60 $B n0=$n n1=$n L=2 mode=each reps=200 warm=50 opt=WL_LDS2D:0,WL_M2D_MAX:$n
done
for n in 128 64 32; do timeout 60 $B n0=$n n1=$n L=0 mode=each reps=200 warm=50; done
} > $O/bench.log 2>&1
for cfg in "1 WL_LDS_MODE:0" "2 WL_LDS_MODE:0,WL_LDS_W:2" "2 WL_LDS_MODE:1"; do
  set -- $cfg
  $R/tools/rp.sh $O/pmc_L$1_$(echo $2 | tr ':,' '__') sq "--kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" $B L=$1 reps=20 warm=5 check=0 opt=$2
  $R/tools/rp.sh $O/pmc_L$1_$(echo $2 | tr ':,' '__') sq2 "--kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" $B L=$1 reps=20 warm=5 check=0 opt=$2
  $R/tools/rp.sh $O/pmc_L$1_$(echo $2 | tr ':,' '__') fetch "--kernel-trace --pmc FETCH_SIZE" $B L=$1 reps=20 warm=5 check=0 opt=$2
  $R/tools/rp.sh $O/pmc_L$1_$(echo $2 | tr ':,' '__') write "--kernel-trace --pmc WRITE_SIZE" $B L=$1 reps=20 warm=5 check=0 opt=$2
done
$R/tools/rp.sh $O/stats_L13 st "--kernel-trace --stats" $B L=13 reps=200 warm=50 check=0 opt=WL_LDS_MODE:1
find $O -name "*.csv" -size +3M -delete
du -sh $O
