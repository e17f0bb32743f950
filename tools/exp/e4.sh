#!/bin/bash
R=$PWD; O=$R/gpurun_out/e4; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
{
for n in 2048 1024; do
timeout 60 $B n0=$n n1=$n L=1 mode=seq reps=500 warm=300 check=1
for mode in 0 1 2; do for w in 1 2 4; do for tj in 16 32 64 128; do
timeout 60 $B n0=$n n1=$n L=1 mode=seq reps=500 warm=300 check=1 opt=WL_TILE:0,WL_M2D_MAX:64,WL_LDS_MODE:$mode,WL_LDS_W:$w,WL_TJ:$tj,WL_WAVES_PER_CU:0,WL_WAVES_MIN:0
done; done; done; done
} > $O/bench.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/e4/bench.log'):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['n'][0], d['filt'], d['L'], d['avg_us'], d['kernel'], d['opt'], d['sum'])
PY
