#!/bin/bash
R=$PWD; O=$R/gpurun_out/full; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
for L in 13 4 2 1; do timeout 60 $B L=$L mode=seq reps=300 warm=300 check=0 | cut -c1-250; done
