#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats and the two HBM-traffic
# PMC passes for the bench command, plus a calibration pass on the plain copy kernel.
# Everything lands under gpurun_out/prof_<tag>/; tools/summarize_prof.py turns it into profiles/.
set -u
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write.log 2>&1
# calibration: a copy of known size through dpx_debug_copy (1 GiB read, 1 GiB written)
CAL="python $REPO/tools/sweep.py --iters 5 --variants 4 --geoms 256x1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cal_fetch -o cal -- $CAL > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cal_write -o cal -- $CAL > $OUT/cal_write.log 2>&1
find $OUT -name "*.csv" | head -40
du -sh $OUT
