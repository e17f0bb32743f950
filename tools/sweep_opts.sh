#!/bin/bash
# option sweeps through tools/wlbench.bin (per-context options, no rebuild): bash tools/sweep_opts.sh   (GPU box, repo root)
B=./tools/wlbench.bin
run() { echo -n "$* : "; for r in 1 2 3; do $B "$@" mode=seq reps=200 warm=60 rot=3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['avg_us'], end=' ')"; done; echo; }
for L in 1 2 3 13; do
  run n0=8192 n1=8192 L=$L filt=sym8
  run n0=8192 n1=8192 L=$L filt=sym8 opt=WL_LONG_BIG_W2:0
done
