#!/bin/bash
# Collect the round's measurement evidence on the GPU box (run through gpurun from the repo root):
#   bench lines for every BASELINE config, rocprofv3 kernel stats of the same commands, the PMC traffic passes of the
#   headline kernel (FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only) and the perf matrix.
#   tools/rp.sh returns as soon as rocprofv3's CSVs are on disk (rocprofv3 does not exit on its own on this image).
R=${1:-r06}
REPO=$PWD
O=$REPO/gpurun_out/prof_$R
mkdir -p $O
export LD_LIBRARY_PATH=$REPO/wavelets.jl_amd:$LD_LIBRARY_PATH
B=$REPO/tools/wlbench.bin
for c in c3 c1 c2 c4 c5; do
  extra="--no-secondary --no-c5"; [ $c = c3 ] && extra=""
  timeout 400 python $REPO/bench.py --workload $c $extra > $O/bench_$c.json 2> $O/bench_$c.err
done
# rocprofv3 --kernel-trace --stats of the bench command itself (short: the statistics need a few hundred launches, not more)
for c in c3 c2 c4 c5; do
  $REPO/tools/rp.sh $O/stats_$c $R "--kernel-trace --stats" python $REPO/bench.py --workload $c --no-cpu --no-secondary --no-c5 --no-pipelined --steps 50 --warmup 3
done
for k in idwt2d idwt2d_sym8 idwt2d_sym5 idwt2d_f64 lift2d lift2d_inv lift3d dwt3d modwt denoise dwt2d_f64 dwt2d_db8 wpt batch2d; do
  $REPO/tools/rp.sh $O/stats_$k $R "--kernel-trace --stats" python $REPO/tools/run_case.py $k 20
done
# PMC: the first launch of the headline transform (an L = 2 call = exactly that kernel: levels 1-2 fused), torch-free harness
for pmc in FETCH_SIZE WRITE_SIZE; do
  $REPO/tools/rp.sh $O/pmc_$pmc $R "--kernel-trace --pmc $pmc" $B L=2 reps=30 warm=5 check=0
done
# ... and of the dominant kernels of C2 / C5 (k_fwd1d_multi, first pass = an L = 4 call) and C4 (k_lift1d_fwd3, levels 1-3 = an L = 3 call)
for pmc in FETCH_SIZE WRITE_SIZE; do
  $REPO/tools/rp.sh $O/pmc_c2_$pmc $R "--kernel-trace --pmc $pmc" $B n0=16777216 n1=1 L=4 reps=30 warm=5 check=0
  $REPO/tools/rp.sh $O/pmc_c5_$pmc $R "--kernel-trace --pmc $pmc" $B dwtc=1 n0=65536 n1=8192 L=4 reps=12 warm=3 check=0
  $REPO/tools/rp.sh $O/pmc_c4_$pmc $R "--kernel-trace --pmc $pmc" python $REPO/tools/run_case.py lift1d_l3 20
done
$REPO/tools/rp.sh $O/pmc_sq $R "--kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" $B L=2 reps=30 warm=5 check=0
$REPO/tools/rp.sh $O/pmc_tcc $R "--kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $B L=2 reps=30 warm=5 check=0
timeout 600 python $REPO/tools/perf_matrix.py > $O/perf_matrix.md 2> $O/perf_matrix.err
timeout 200 python $REPO/tools/time_wpt.py > $O/wpt_timings.md 2>/dev/null
timeout 200 python $REPO/tools/time_batch.py > $O/batch_of_images.md 2>/dev/null
$REPO/tools/wlbench_mgpu.bin gpus=1 steps=10 > $O/native_mgpu_1rank.json 2> $O/native_mgpu_1rank.err
find $O -name "*.csv" -size +4M -delete
find $O -name "*.csv" | head -60
