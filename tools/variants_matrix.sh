#!/bin/bash
# bash tools/variants_matrix.sh name1 [name2 ...] : the variants against the product build on a matrix of forward 2-D cases
for args in "n0=8192 n1=8192 L=1" "n0=8192 n1=8192 L=13 filt=sym5" "n0=8192 n1=8192 L=13 filt=sym8" "n0=8192 n1=8192 L=13 dtype=f64" \
            "n0=8192 n1=8192 L=13 filt=haar" "n0=4096 n1=4096 L=12" "n0=2048 n1=2048 L=11" "n0=1024 n1=1024 L=10" "n0=16384 n1=4096 L=12" "n0=4096 n1=16384 L=12"; do
  echo "== $args"
  bash tools/variants.sh "$args mode=seq reps=200 warm=60 rot=3" "$@" | sort -k1,1 -k2,2n | awk '{a[$1]=a[$1]" "$2; k[$1]=$3} END{for(n in a) print n, a[n], k[n]}' | sort
done
