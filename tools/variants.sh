#!/bin/bash
# A/B of build variants of the library (tools/alt/<name>/libwavelets_mi355x.so) against the product build with tools/wlbench.bin.
#   bash tools/variants.sh "<wlbench args>" name1 name2 ...      (run on the GPU box from the repo root; interleaved, 3 rounds)
R=$PWD; B=$R/tools/wlbench.bin; ARGS=$1; shift
for rep in 1 2 3; do
  for v in base "$@"; do
    if [ $v = base ]; then export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib; else export LD_LIBRARY_PATH=$R/tools/alt/$v:/opt/rocm/lib; fi
    echo -n "$v: "; timeout 120 $B $ARGS | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['avg_us'], d['kernel'], d['sum'])"
  done
done
