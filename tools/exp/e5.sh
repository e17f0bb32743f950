#!/bin/bash
R=$PWD; O=$R/gpurun_out/e5; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
{
for rep in 1 2; do
for f in db4 sym5 haar; do
timeout 60 $B L=2 filt=$f mode=seq reps=300 warm=300 check=0
timeout 60 $B L=2 filt=$f mode=seq reps=300 warm=300 check=0 opt=WL_PAIR_SAFEWAIT:1
done; done
} > $O/bench.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/e5/bench.log'):
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['n'][0], d['filt'], d['L'], d['avg_us'], d['kernel'], d['opt'])
PY
