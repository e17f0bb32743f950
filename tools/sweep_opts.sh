#!/bin/bash
# Option sweeps through tools/wlbench.bin (per-context options: no rebuild).  On the GPU box, from the repo root:
#   bash tools/sweep_opts.sh "<wlbench args>" OPT_A:1,OPT_B:2 OPT_A:0 ...     one line per option set (and one without options),
#   three runs each, 200 back-to-back calls on three rotating inputs.  Example (round 5, the long-filter inverse tiles):
#   bash tools/sweep_opts.sh "n0=8192 n1=8192 L=13 fw=0 filt=sym8" WL_TILE_INV_LONG:0
B=./tools/wlbench.bin
ARGS=$1; shift
run() { echo -n "${2:-(default)} : "; for r in 1 2 3; do $B $1 mode=seq reps=200 warm=60 rot=3 ${2:+opt=$2} | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['avg_us'], end=' ')"; done; echo; }
run "$ARGS"
for o in "$@"; do run "$ARGS" "$o"; done
