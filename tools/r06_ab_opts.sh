#!/bin/bash
# bash tools/r06_ab_opts.sh "<wlbench args>" "opt1" "opt2" ...   interleaved rounds of per-context option sets on the product build ("-" = none)
R=$PWD; B=$R/tools/wlbench.bin; ARGS=$1; shift
export LD_LIBRARY_PATH=${WLLIB:-$R/wavelets.jl_amd}:/opt/rocm/lib
for rep in 1 2 3 4; do
  for o in "$@"; do
    oo=""; [ "$o" != "-" ] && oo="opt=$o"
    echo -n "$o: "; timeout 120 $B $ARGS $oo | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['avg_us'])"
  done
done | sort -k1,1 -k2,2n | awk '{a[$1]=a[$1]" "$2} END{for(n in a) print n, a[n]}' | sort
