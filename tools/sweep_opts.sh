#!/bin/bash
# option sweeps through tools/wlbench.bin (per-context options, no rebuild): bash tools/sweep_opts.sh   (GPU box, repo root)
B=./tools/wlbench.bin
run() { echo -n "$* : "; for r in 1 2 3; do $B "$@" mode=seq reps=200 warm=60 rot=3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['avg_us'], end=' ')"; done; echo; }
for f in db6 sym8 db10; do
  for v in 1 0; do run n0=8192 n1=8192 L=13 fw=0 filt=$f opt=WL_TILE_INV_LONG:$v; done
done
for v in 1 0; do run n0=8192 n1=8192 L=13 fw=0 filt=sym8 dtype=f64 opt=WL_TILE_INV_LONG:$v; done
for v in 1 0; do run n0=1024 n1=1024 L=10 fw=0 filt=sym8 opt=WL_TILE_INV_LONG:$v; done
for v in 1 0; do run n0=512 n1=512 L=9 fw=0 filt=db6 opt=WL_TILE_INV_LONG:$v; done
