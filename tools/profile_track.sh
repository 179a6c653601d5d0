#!/bin/bash
# rocprofv3 kernel-trace stats + HBM traffic counters for the secondary track workload (tile kernel).
set -u
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_track_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --workload track --steps 10 --warmup 2"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_l2 -o bench -- $CMD > $OUT/pmc_l2.log 2>&1
du -sh $OUT
