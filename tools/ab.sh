#!/bin/bash
# bash tools/ab.sh "<wlbench args>" variant...  : one summary line per build (three interleaved rounds, sorted times)
bash tools/variants.sh "$1 mode=seq reps=200 warm=60 rot=3" "${@:2}" | sort -k1,1 -k2,2n | awk '{a[$1]=a[$1]" "$2; k[$1]=$3} END{for(n in a) print n, a[n], k[n]}' | sort
