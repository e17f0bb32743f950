#!/bin/bash
# Counters for the dominant kernel of every secondary configuration that sits below 0.35 of the HBM roofline (round-5 review item 3):
# four rocprofv3 passes each (--kernel-trace only, never with a trace domain): FETCH_SIZE | WRITE_SIZE | SQ group 1 | SQ group 2.
# Run on the GPU box from the repo root: bash tools/r06_secondary_pmc.sh [outdir-name];  python tools/secondary_pmc.py DIR -> markdown + json
R=$PWD; O=$R/gpurun_out/${1:-secpmc}; mkdir -p $O
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:/opt/rocm/lib
B=$R/tools/wlbench.bin
G1="FETCH_SIZE"
G2="WRITE_SIZE"
G3="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"
G4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU"
run4() {   # name cmd...
  nm=$1; shift
  i=1
  for g in "$G1" "$G2" "$G3" "$G4"; do
    $R/tools/rp.sh $O/${nm}_g$i r06 "--kernel-trace --pmc $g" "$@" > /dev/null 2>&1
    i=$((i+1))
  done
  echo "done $nm"
}
run4 dwt3d       python $R/tools/run_case.py dwt3d 6
run4 idwt3d      python $R/tools/run_case.py idwt3d 6
run4 lift2d      python $R/tools/run_case.py lift2d 8
run4 lift2d_inv  python $R/tools/run_case.py lift2d_inv 8
run4 sym8_fwd    $B filt=sym8 L=13 rot=2 reps=8 warm=2 check=0
run4 sym8_inv    $B filt=sym8 L=13 fw=0 rot=2 reps=8 warm=2 check=0
run4 modwt       python $R/tools/run_case.py modwt 2
run4 batt6       python $R/tools/run_case.py batt6 4
run4 c2          $B n0=16777216 n1=1 L=24 rot=3 reps=12 warm=3 check=0
run4 c4          python $R/tools/run_case.py c4 12
run4 c3          $B L=13 rot=3 reps=10 warm=3 check=0
find $O -name "*.csv" -size +6M -delete
ls $O | head -50
