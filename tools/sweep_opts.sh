#!/bin/bash
# option sweeps through tools/wlbench.bin (per-context options, no rebuild): bash tools/sweep_opts.sh   (GPU box, repo root)
B=./tools/wlbench.bin
run() { echo -n "$* : "; for r in 1 2 3; do $B "$@" mode=seq reps=200 warm=60 rot=3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['avg_us'], end=' ')"; done; echo; }
for f in db6 sym8 db10; do
  for mx in 1024 2048 4096; do
    run n0=8192 n1=8192 L=13 filt=$f opt=WL_TILE_LONG_MAX:$mx
  done
done
for mx in 1024 2048; do run n0=2048 n1=2048 L=11 filt=sym8 opt=WL_TILE_LONG_MAX:$mx; done
for t in 256 512; do run n0=1024 n1=1024 L=10 filt=sym8 opt=WL_TILE_LONG_MAX:1024,WL_TILE_THREADS:$t; done
