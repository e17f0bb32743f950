#!/bin/bash
R=$PWD; O=$R/gpurun_out/e2; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_exchange_2d_kernel or full_size_elementwise" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
{
for n in 8192 4096 2048 1024; do
timeout 60 $B n0=$n n1=$n L=2 mode=seq reps=300 warm=300 check=1
for w in 2 4; do for tj in 128 64 32; do
timeout 60 $B n0=$n n1=$n L=2 mode=seq reps=300 warm=300 check=1 opt=WL_LDS_PAIR_MIN:0,WL_PAIR_W:$w,WL_TJ2:$tj
done; done; done
for f in haar db2 db3 sym5; do
timeout 60 $B L=2 filt=$f mode=seq reps=300 warm=300 check=1
timeout 60 $B L=2 filt=$f mode=seq reps=300 warm=300 check=1 opt=WL_LDS_PAIR_MIN:0,WL_PAIR_W:2
done
timeout 60 $B L=13 mode=seq reps=300 warm=300 check=1
timeout 60 $B L=13 mode=seq reps=300 warm=300 check=1 opt=WL_LDS_PAIR_MIN:0,WL_PAIR_W:2
timeout 60 $B L=4 mode=seq reps=300 warm=300 check=1 opt=WL_LDS_PAIR_MIN:0,WL_PAIR_W:2
timeout 60 $B L=4 mode=seq reps=300 warm=300 check=1 opt=WL_LDS_PAIR_MIN:4194304,WL_PAIR_W:2
} > $O/bench.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/e2/bench.log'):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['n'][0], d['filt'], d['L'], d['avg_us'], d['kernel'], d['opt'], d['sum'])
PY
