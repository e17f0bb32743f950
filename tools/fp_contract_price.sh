#!/bin/bash
# Price of bit-exactness: the product build (-ffp-contract=off) against the opt-in fused build of the same sources
# (libwavelets_mi355x_fma.so, `make -C wavelets.jl_amd/csrc FMA=1`).  Run on the GPU box from the repo root:
#   bash tools/fp_contract_price.sh          -> gpurun_out/fpc/{wlbench.log, matrix_exact.md, matrix_fused.md}
# tools/fp_contract_table.py turns the three files into profiles/r05_fp_contract_price.md.
O=$PWD/gpurun_out/fpc; mkdir -p $O
R=$PWD
B=$R/tools/wlbench.bin
mkdir -p /tmp/wl_alt && cp $R/wavelets.jl_amd/libwavelets_mi355x_fma.so /tmp/wl_alt/libwavelets_mi355x.so
{
for lib in exact fused; do
  if [ $lib = exact ]; then export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib; else export LD_LIBRARY_PATH=/tmp/wl_alt:/opt/rocm/lib; fi
  for rep in 1 2; do
  echo "=== $lib"
  timeout 60 $B n0=8192 n1=8192 L=1 mode=seq reps=300 warm=100
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=300 warm=100
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=300 warm=100 fw=0
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=300 warm=100 filt=sym5
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=300 warm=100 filt=sym5 fw=0
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=200 warm=50 filt=sym8
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=200 warm=50 filt=sym8 fw=0
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=200 warm=50 dtype=f64
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=200 warm=50 dtype=f64 fw=0
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=200 warm=50 filt=cdf97lift
  timeout 60 $B n0=8192 n1=8192 L=13 mode=seq reps=200 warm=50 filt=cdf97lift fw=0
  timeout 60 $B n0=512 n1=512 n2=512 L=9 mode=seq reps=200 warm=50
  timeout 60 $B n0=512 n1=512 n2=512 L=9 mode=seq reps=200 warm=50 fw=0
  timeout 60 $B n0=16777216 n1=1 L=24 mode=seq reps=300 warm=100
  timeout 60 $B n0=16777216 n1=1 L=24 mode=seq reps=300 warm=100 fw=0
  timeout 60 $B n0=16777216 n1=1 L=24 mode=seq reps=300 warm=100 filt=cdf97lift
  timeout 60 $B n0=16777216 n1=1 L=24 mode=seq reps=300 warm=100 filt=cdf97lift fw=0
  timeout 60 $B n0=65536 n1=8192 L=16 dwtc=1 mode=seq reps=50 warm=10
  done
done
} > $O/wlbench.log 2>&1
timeout 600 python tools/perf_matrix.py --arithmetic exact > $O/matrix_exact.md 2> $O/matrix_exact.err
timeout 600 python tools/perf_matrix.py --arithmetic fused > $O/matrix_fused.md 2> $O/matrix_fused.err
tail -3 $O/wlbench.log; tail -2 $O/matrix_fused.md
