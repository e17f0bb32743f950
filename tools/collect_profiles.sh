#!/bin/bash
# Collect the round's measurement evidence on the GPU box (run through gpurun from the repo root):
#   bench lines for every BASELINE config, rocprofv3 kernel stats, and the PMC traffic passes of the headline
#   kernel (FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only).  rocprofv3 does not exit on
#   its own on this image once the child has finished, hence the hard timeouts; the CSVs are complete by then.
R=${1:-r01}
O=$PWD/gpurun_out/prof_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
REPO=$OLDPWD
for c in c3 c1 c2 c4 c5; do
  extra="--no-secondary"; [ $c = c3 ] && extra=""
  timeout 280 python $REPO/bench.py --workload $c $extra > $O/bench_$c.json 2> $O/bench_$c.err
done
for c in c3 c2 c4 c5; do
  timeout -s KILL 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -o $R -- \
    python $REPO/bench.py --workload $c --no-cpu --no-secondary --steps 50 --warmup 3 > $O/stats_$c.log 2>&1
done
for k in idwt2d lift2d dwt3d modwt; do
  timeout -s KILL 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$k -o $R -- \
    python $REPO/tools/run_case.py $k 20 > $O/stats_$k.log 2>&1
done
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 100 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_$pmc -o $R -- \
    python $REPO/bench.py --workload c3 --no-cpu --no-secondary --steps 30 --warmup 0 --levels 2 > $O/pmc_$pmc.log 2>&1
done
find $O -name "*.csv" | head -50
