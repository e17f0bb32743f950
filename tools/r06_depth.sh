#!/bin/bash
# What each launch of the 8192^2 chain costs un-profiled: back-to-back transforms at increasing depth (three rotating inputs).
R=$PWD; B=$R/tools/wlbench.bin
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
for round in 1 2 3; do
for L in 1 2 4 7 13; do
  echo -n "L=$L "; $B n0=8192 n1=8192 L=$L rot=3 reps=200 warm=60 check=0 ${@} | grep -o '"us_per_call": [0-9.]*\|"avg_us": [0-9.]*' | head -2 | tr '\n' ' '; echo
done; done
for cfg in "n0=2048 n1=2048 L=2" "n0=2048 n1=2048 L=11" "n0=512 n1=512 L=3" "n0=512 n1=512 L=9" "n0=64 n1=64 L=6"; do
  echo -n "$cfg "; $B $cfg rot=3 reps=300 warm=60 check=0 ${@} | grep -o '"us_per_call": [0-9.]*\|"avg_us": [0-9.]*' | head -2 | tr '\n' ' '; echo
done
