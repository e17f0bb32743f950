#!/bin/bash
# rp.sh OUTDIR NAME "<rocprofv3 options>" cmd...  -- run rocprofv3 and return as soon as its CSVs are on disk.
# (rocprofv3 does not exit on its own on this image once the child has finished; the CSVs are complete long before.)
OUT=$1; NAME=$2; OPTS=$3; shift 3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
setsid rocprofv3 $OPTS --output-format csv -d "$OUT" -o "$NAME" -- "$@" > "$OUT/$NAME.log" 2>&1 &
PID=$!
last=-1; stable=0
for i in $(seq 1 ${RP_MAX_ITERS:-240}); do
  sleep 0.5
  if ! kill -0 $PID 2>/dev/null; then break; fi
  sz=$(find "$OUT" -name "${NAME}_*.csv" -printf "%s+" 2>/dev/null)
  if [ -n "$sz" ] && [ "$sz" = "$last" ]; then stable=$((stable+1)); else stable=0; fi
  last=$sz
  if [ $stable -ge 6 ]; then break; fi
done
kill -KILL -- -$PID 2>/dev/null
wait $PID 2>/dev/null
ls "$OUT" | grep "^$NAME" | head -5
