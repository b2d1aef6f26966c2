#!/bin/bash
# Standard measurement set on the GPU box (run through gpurun from the repo root):
#   scripts/gpu_profile.sh TAG   -> gpurun_out/TAG_bench.json, TAG_stats/ (rocprofv3 kernel stats of the default bench),
#                                   TAG_pmc_{fetch,write,sq}/ (three separate counter passes), TAG_pmc_summary.json
set -e -o pipefail
TAG=${1:-run}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --no-cpu-baseline > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 scripts/profile_kernels.py > $OUT/${TAG}_pmc_sq.log 2>&1
python3 scripts/pmc_summary.py $OUT/${TAG}_pmc_summary.json $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_sq > $OUT/${TAG}_pmc_summary.log
find $OUT/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
echo "profile set $TAG done"
