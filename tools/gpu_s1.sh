#!/bin/bash
# GPU session 1 (round 2): diagnostics of the headline kernels before redesign.  Run from the repo root via gpurun.
O=$PWD/gpurun_out/s1; mkdir -p $O
R=$PWD
export LD_LIBRARY_PATH=$R/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$R/tools/wlbench.bin
{
echo "== valu probe"; timeout 120 $R/tools/probes/valu_probe.bin
echo "== march probe"; timeout 200 $R/tools/probes/march_probe.bin
echo "== wlbench baselines"
for args in "L=13" "L=2 mode=each" "L=1 mode=each" "L=13 fw=0" "L=1 fw=0 mode=each" \
            "n0=8256 n1=8192 L=1 mode=each" "n0=8192 n1=8256 L=1 mode=each" "n0=8256 n1=8192 L=2 mode=each" "n0=8320 n1=8192 L=1 mode=each" \
            "n0=4096 n1=4096 L=12" "n0=2048 n1=2048 L=11" "n0=2048 n1=2048 L=1 mode=each" "n0=1024 n1=1024 L=1 mode=each" "n0=512 n1=512 L=1 mode=each" \
            "n0=2048 n1=2048 L=2 mode=each" "n0=16777216 n1=1 L=24" "n0=16777216 n1=1 L=24 fw=0" "filt=sym5 L=13" "filt=db2 L=13" "dtype=f64 L=13"; do
  timeout 60 $B $args reps=100 warm=30
done
} > $O/probes.log 2>&1
# PMC passes on the pair kernel (L=2) and the single-level kernel (L=1)
for L in 2 1; do
  $R/tools/rp.sh $O/pmc_L$L sq "--kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" $B L=$L reps=20 warm=5 check=0
  $R/tools/rp.sh $O/pmc_L$L tcc1 "--kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $B L=$L reps=20 warm=5 check=0
  $R/tools/rp.sh $O/pmc_L$L tcc2 "--kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" $B L=$L reps=20 warm=5 check=0
  $R/tools/rp.sh $O/pmc_L$L tcp "--kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TA_BUSY_avr" $B L=$L reps=20 warm=5 check=0
  $R/tools/rp.sh $O/pmc_L$L stall "--kernel-trace --pmc MemUnitStalled WriteUnitStalled" $B L=$L reps=20 warm=5 check=0
  $R/tools/rp.sh $O/pmc_L$L fetch "--kernel-trace --pmc FETCH_SIZE" $B L=$L reps=20 warm=5 check=0
  $R/tools/rp.sh $O/pmc_L$L write "--kernel-trace --pmc WRITE_SIZE" $B L=$L reps=20 warm=5 check=0
done
$R/tools/rp.sh $O/stats_L13 st "--kernel-trace --stats" $B L=13 reps=200 warm=50 check=0
# keep only small files
find $O -name "*.csv" -size +3M -delete
du -sh $O; tail -5 $O/probes.log
