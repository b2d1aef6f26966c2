#!/bin/bash
# The measurement set behind every number in DESIGN.md (run through gpurun from the repo root):
#   scripts/gpu_profile.sh TAG   ->  gpurun_out/TAG_*  ; then  python3 scripts/store_profiles.py TAG roundN  copies the summaries to profiles/roundN/
#   bench line (with end_to_end + roofline), rocprofv3 kernel stats of the SAME command, four separate counter passes over a fixed launch
#   sequence (FETCH_SIZE; WRITE_SIZE; SQ busy / wait; FP64 instruction mix), the frame-sharded tick with a one-rank RCCL communicator,
#   the user-level call's wall-clock breakdown, the other shapes DESIGN quotes, kernel stats of the config-5 shard (24 x 6 250 x 200).
set -e -o pipefail
TAG=${1:-run}
PART=${2:-all}   # a gpurun call is limited to 20 minutes: `scripts/gpu_profile.sh TAG a`, then `... TAG b`, then `... TAG c` (all = everything in one go; d = the calibrate() section of c alone)
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$PART" = "all" ] || [ "$PART" = "a" ]; then
python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench done"
# (--no-other-configs: the headline tick alone -- configs[0] runs the same k_syrk / k_solve_backsub instances at a tenth of the size and would be averaged in)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --no-cpu-baseline --no-end-to-end --no-other-configs > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.err
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_write.log 2>&1
echo "hbm counters done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${TAG}_pmc_f64 -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_f64.log 2>&1
echo "sq counters done"
python3 scripts/pmc_summary.py $OUT/${TAG}_pmc_summary.json $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_f64 > $OUT/${TAG}_pmc_summary.log
find $OUT/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
MCBA_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_dist_stats -- python3 bench.py --no-cpu-baseline --no-end-to-end > $OUT/${TAG}_dist_bench.json 2> $OUT/${TAG}_dist.err
find $OUT/${TAG}_dist_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_dist_kernel_stats.csv \;
echo "forced-dist done"
fi
if [ "$PART" = "all" ] || [ "$PART" = "b" ]; then
python3 scripts/e2e_breakdown.py > $OUT/${TAG}_e2e_breakdown.json 2> $OUT/${TAG}_e2e.err
# round 5: the user-level call where users are (the reference tutorial's recording, BASELINE configs[0]), its kernels by rocprofv3, and the
# timelines of the small-shape ticks (what a kernel boundary costs in the stream)
MCBA_E2E_SHAPE=6,2130,5,7 python3 scripts/e2e_breakdown.py 2130 9 > $OUT/${TAG}_e2e_tutorial.json 2>> $OUT/${TAG}_e2e.err
MCBA_E2E_SHAPE=2,50,6,9 python3 scripts/e2e_breakdown.py 50 9 > $OUT/${TAG}_e2e_config0.json 2>> $OUT/${TAG}_e2e.err
MCBA_E2E_MISSING=0.05 python3 scripts/e2e_breakdown.py > $OUT/${TAG}_e2e_missing.json 2>> $OUT/${TAG}_e2e.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_e2e_stats -- python3 scripts/e2e_breakdown.py 10000 5 > /dev/null 2>> $OUT/${TAG}_e2e.err
find $OUT/${TAG}_e2e_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_e2e_kernel_stats.csv \;
for S in "config1:6,1000,6,9,1" "config0:2,50,6,9" "tutorial:6,2130,5,7"; do
  N=${S%%:*}; SH=${S#*:}
  MCBA_SHAPES="$SH" rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_tl_$N -- python3 scripts/other_shapes.py > /dev/null 2>> $OUT/${TAG}_e2e.err
  find $OUT/${TAG}_tl_$N -name "*kernel_trace.csv" -exec cp {} $OUT/${TAG}_tl_${N}_trace.csv \;
  python3 scripts/tick_timeline.py $OUT/${TAG}_tl_${N}_trace.csv > $OUT/${TAG}_tick_timeline_$N.txt
  rm -rf $OUT/${TAG}_tl_$N $OUT/${TAG}_tl_${N}_trace.csv
done
echo "round-5 e2e done"
python3 scripts/other_shapes.py > $OUT/${TAG}_other_shapes.json 2> $OUT/${TAG}_shapes.err
MCBA_SHAPES="24,6250,10,20" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_shard5_stats -- python3 scripts/other_shapes.py > $OUT/${TAG}_shard5.json 2> $OUT/${TAG}_shard5.err
find $OUT/${TAG}_shard5_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_shard5_kernel_stats.csv \;
fi
if [ "$PART" = "all" ] || [ "$PART" = "c" ]; then
# round 4: the shapes that are not one round of the wavefront slots -- rocprofv3 kernel stats of the LM loop at BASELINE configs[1]
# (6 x 1 000 x 54, intrinsics held fixed: the 6-wide camera block), the same size with every parameter free, configs[0] (2 x 50), a
# configs[3] shard (6 x 12 500: fused round + point-split tail); HBM counter passes at configs[1]; the FULL configs[4] problem on one GPU
for S in "config1:6,1000,6,9,1" "c1free:6,1000,6,9" "config0:2,50,6,9" "shard4:6,12500,6,9"; do
  N=${S%%:*}; SH=${S#*:}
  MCBA_SHAPES="$SH" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${N}_stats -- python3 scripts/other_shapes.py > $OUT/${TAG}_${N}.json 2> $OUT/${TAG}_${N}.err
  find $OUT/${TAG}_${N}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_${N}_kernel_stats.csv \;
done
MCBA_SHAPE="6,1000,6,9" MCBA_FIXED=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_c1_fetch -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_c1_fetch.log 2>&1
MCBA_SHAPE="6,1000,6,9" MCBA_FIXED=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_c1_write -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_c1_write.log 2>&1
MCBA_SHAPE="6,1000,6,9" MCBA_FIXED=1 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}_pmc_c1_sq -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_c1_sq.log 2>&1
python3 scripts/pmc_summary.py $OUT/${TAG}_pmc_config1_summary.json $OUT/${TAG}_pmc_c1_fetch $OUT/${TAG}_pmc_c1_write $OUT/${TAG}_pmc_c1_sq > $OUT/${TAG}_pmc_config1_summary.log
echo "round-4 shapes done"
MCBA_SHAPES="24,50000,10,20" python3 scripts/other_shapes.py > $OUT/${TAG}_config5_full.json 2> $OUT/${TAG}_config5_full.err
fi
if [ "$PART" = "all" ] || [ "$PART" = "c" ] || [ "$PART" = "d" ]; then
# round 6: calibrate() -- wall time with its stages at the tutorial shape and at 6 x 10 000 x 54, rocprofv3 kernel stats of the call and of the dense
# kernels' fixed launch sequence, HBM counter passes of the latter
python3 scripts/calibrate_time.py > $OUT/${TAG}_calibrate.json 2> $OUT/${TAG}_calibrate.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cal_stats -- python3 scripts/calibrate_time.py 6,10000,6,9 > /dev/null 2>> $OUT/${TAG}_calibrate.err
find $OUT/${TAG}_cal_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_calibrate_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_calk_stats -- python3 scripts/profile_calibrate.py > $OUT/${TAG}_calibrate_kernels.json 2>> $OUT/${TAG}_calibrate.err
find $OUT/${TAG}_calk_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_calibrate_dense_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_cal_fetch -- python3 scripts/profile_calibrate.py > $OUT/${TAG}_pmc_cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_cal_write -- python3 scripts/profile_calibrate.py > $OUT/${TAG}_pmc_cal_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d $OUT/${TAG}_pmc_cal_sq -- python3 scripts/profile_calibrate.py > $OUT/${TAG}_pmc_cal_sq.log 2>&1
python3 scripts/pmc_summary.py $OUT/${TAG}_pmc_calibrate_summary.json $OUT/${TAG}_pmc_cal_fetch $OUT/${TAG}_pmc_cal_write $OUT/${TAG}_pmc_cal_sq > $OUT/${TAG}_pmc_calibrate_summary.log
rm -rf $OUT/${TAG}_cal_stats $OUT/${TAG}_calk_stats
echo "round-6 calibrate done"
fi
echo "profile set $TAG done"
