#!/bin/bash
# Extra PMC passes for the headline kernel: SQ occupancy/issue counters and L2 hit rate (separate passes).
set -u
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_ctr_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 3 --no-cpu"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace -d $OUT/sq1 -o b -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq2 -o b -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $OUT/l2 -o b -- $CMD > $OUT/l2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d $OUT/l1 -o b -- $CMD > $OUT/l1.log 2>&1
du -sh $OUT; grep -il "error\|invalid" $OUT/*.log | head
