#!/bin/bash
# Round-5 review item 2: does less halo traffic make the first launch of the headline faster?  Chunk length / residency variants of
# k_fwd2d_pair (8192^2 db4 f32, L = 2 = that launch alone): time (three interleaved rounds, 3 rotating inputs) and FETCH / WRITE traffic.
R=$PWD; O=$R/gpurun_out/${1:-halo}; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
VARS=("-" "WL_TJ2:64" "WL_TJ2:256,WL_PAIR_WG_PER_CU:2" "WL_TJ2:512,WL_PAIR_WG_PER_CU:1" "WL_PAIR_W:4" "WL_PAIR_W:4,WL_TJ2:256,WL_PAIR_WG_PER_CU:1")
bash tools/r06_ab_opts.sh "L=2 rot=3 reps=200 warm=60 check=0" "${VARS[@]}" > $O/times.txt
cat $O/times.txt
i=0
for v in "${VARS[@]}"; do
  oo=""; [ "$v" != "-" ] && oo="opt=$v"
  for pmc in FETCH_SIZE WRITE_SIZE; do
    $R/tools/rp.sh $O/v${i}_$pmc r06 "--kernel-trace --pmc $pmc" $B L=2 rot=3 reps=12 warm=3 check=0 $oo > /dev/null 2>&1
  done
  python3 - "$O" $i "$v" <<'PY'
import csv, glob, sys
O, i, v = sys.argv[1], sys.argv[2], sys.argv[3]
vals = {}
for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = []
    for h in glob.glob(f"{O}/v{i}_{pmc}/**/*counter_collection.csv", recursive=True):
        xs += [float(r["Counter_Value"]) for r in csv.DictReader(open(h)) if "k_fwd2d_pair" in r["Kernel_Name"] and r["Counter_Name"] == pmc]
    vals[pmc] = sum(xs) / len(xs) if xs else float("nan")
t = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
print(f"{v}: fetch {2*vals['FETCH_SIZE']*1024/1e6:.1f} MB write {vals['WRITE_SIZE']*1024/1e6:.1f} MB traffic {t/1e6:.1f} MB = {t/536870912:.3f} x algorithmic")
PY
  i=$((i+1))
done | tee $O/traffic.txt
find $O -name "*.csv" -size +2M -delete
