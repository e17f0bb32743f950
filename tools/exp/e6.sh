#!/bin/bash
R=$PWD; O=$R/gpurun_out/e6; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
{
for rep in 1 2; do
for f in db4 sym5; do
timeout 60 $B L=1 filt=$f mode=seq reps=300 warm=300 check=0
timeout 60 $B L=1 filt=$f mode=seq reps=300 warm=300 check=0 opt=WL_LDS_COUNT_STORES:1
done; done
timeout 60 $B L=13 mode=seq reps=300 warm=300 check=0
} > $O/bench.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/e6/bench.log'):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['n'][0], d['filt'], d['L'], d['avg_us'], d['kernel'], d['opt'])
PY
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused_pair_2d or lds_exchange" 2>&1 | grep -E "AssertionError:|passed|failed" | cut -c1-200; done
