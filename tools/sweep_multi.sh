#!/bin/bash
# scratch: sweep tile size / fused levels of k_fwd1d_multi on C5 and C2
for w in c5 c2; do
for ts in 8192 4096 2048; do
for nl in 2 3 4 6; do
  r=$(WL_TS=$ts WL_NLMAX=$nl timeout 120 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])")
  echo "$w TS=$ts NL=$nl ms=$r"
done; done; done
