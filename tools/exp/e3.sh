#!/bin/bash
R=$PWD; O=$R/gpurun_out/e3; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
{
# 2048^2 -> 1 with the existing kernel families
for o in "" "WL_TILE_MAX:2048" "WL_TILE_MAX:2048,WL_TILE_NL3_MAX:2048" "WL_TILE_MAX:2048,WL_TILE_NL3_MAX:512" "WL_TILE_NL3_MAX:1024" "WL_TILE_MAX:2048,WL_TILE_THREADS:512" "WL_TILE_MAX:2048,WL_TILE_THREADS:256"; do
timeout 60 $B n0=2048 n1=2048 L=11 mode=seq reps=500 warm=300 check=1 opt=$o
done
for L in 1 2 3 4 5; do
timeout 60 $B n0=2048 n1=2048 L=$L mode=seq reps=500 warm=300 check=1 opt=WL_TILE_MAX:2048
done
for n in 64 128 256 512 1024; do
timeout 60 $B n0=$n n1=$n mode=seq reps=500 warm=300 check=1
done
timeout 60 $B n0=512 n1=512 mode=seq reps=500 warm=300 check=1 opt=WL_TILE_NL3_MAX:512
timeout 60 $B n0=1024 n1=1024 mode=seq reps=500 warm=300 check=1 opt=WL_TILE_NL3_MAX:1024
} > $O/bench.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/e3/bench.log'):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['n'][0], d['filt'], d['L'], d['avg_us'], d['kernel'], d['opt'], d['sum'])
PY
