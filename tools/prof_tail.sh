#!/bin/bash
# SQ counters + kernel stats of the forward tail kernels of the 8192^2 transform (k_fwd2d_tileB on the 2048^2 block etc.)
# and of the dominant pair kernel.  Run on the GPU box from the repo root:  bash tools/prof_tail.sh [outdir-name]
R=$PWD; O=$R/gpurun_out/${1:-tailprof}; mkdir -p $O
B=$R/tools/wlbench.bin
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
P3="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"
for cfg in "tileB n0=2048 n1=2048 L=2" "pair n0=8192 n1=8192 L=2" "t512 n0=512 n1=512 L=2"; do
  set -- $cfg; nm=$1; shift
  $R/tools/rp.sh $O/${nm}_p1 r05 "--kernel-trace --pmc $P1" $B "$@" reps=20 warm=5 check=0
  $R/tools/rp.sh $O/${nm}_p2 r05 "--kernel-trace --pmc $P2" $B "$@" reps=20 warm=5 check=0
  $R/tools/rp.sh $O/${nm}_p3 r05 "--kernel-trace --pmc $P3" $B "$@" reps=20 warm=5 check=0
done
$R/tools/rp.sh $O/stats_c3 r05 "--kernel-trace --stats" $B n0=8192 n1=8192 L=13 reps=200 warm=50 check=0
find $O -name "*.csv" -size +2M -delete
python3 - <<PY
import csv, glob, collections, os
O="$O"
for d in sorted(glob.glob(O+"/*_p?")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            if "fill" in k: continue
            print(os.path.basename(d), k, {c: round(sum(x)/len(x)) for c,x in v.items()})
for f in glob.glob(O+"/stats_c3/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
PY
