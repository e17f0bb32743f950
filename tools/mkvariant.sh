#!/bin/bash
# tools/mkvariant.sh NAME "EXTRA FLAGS" file1.hip [file2.hip ...]   -> tools/alt/NAME/libwavelets_mi355x.so
# An experiment build: the listed translation units recompiled with the extra flags, every other object taken from the product build
# (make -C wavelets.jl_amd/csrc first).  tools/variants.sh A/Bs the variants on the GPU box.  tools/alt/ is not tracked.
set -e
NAME=$1; FLAGS=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); S=$R/wavelets.jl_amd/csrc; O=$R/tools/alt/$NAME; mkdir -p $O
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -DWL_BUILDING_LIB"
OBJS=""
for f in $S/*.o; do b=$(basename $f .o); skip=0; for t in "$@"; do [ "$b.hip" = "$t" ] && skip=1; done; [ $skip = 0 ] && OBJS="$OBJS $f"; done
for t in "$@"; do
  extra=""; case $t in wl_vlong.hip|wl_inv2d_long.hip|wl_fwd3d.hip|wl_inv3d.hip) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc $BASE $extra $FLAGS -c $S/$t -o $O/${t%.hip}.o &
done
wait
NEW=""; for t in "$@"; do NEW="$NEW $O/${t%.hip}.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libwavelets_mi355x.so $OBJS $NEW
echo built $O/libwavelets_mi355x.so
