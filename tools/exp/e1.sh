#!/bin/bash
R=$PWD; O=$R/gpurun_out/e1; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_exchange_2d_kernel or full_size_elementwise" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
{
timeout 60 $B L=1 mode=seq reps=300 warm=300 check=1
timeout 60 $B L=2 mode=seq reps=300 warm=300 check=1
for st in 0 1 2; do for w in 4 2; do for tj in 128 256 64; do
timeout 60 $B L=2 mode=seq reps=300 warm=300 check=1 opt=WL_LDS_PAIR_MIN:0,WL_PAIR_W:$w,WL_TJ2:$tj,WL_PAIR_ST:$st
done; done; done
timeout 60 $B L=13 mode=seq reps=300 warm=300 check=1
timeout 60 $B L=13 mode=seq reps=300 warm=300 check=1 opt=WL_LDS_PAIR_MIN:0
} > $O/bench.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/e1/bench.log'):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['L'], d['avg_us'], d['kernel'], d['opt'], d['sum'])
PY
