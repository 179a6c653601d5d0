#!/bin/bash
# VALU / LDS counters of the kernels behind the two bench workloads (separate rocprofv3 --pmc passes, kernel trace only).
set -u
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_ctr_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for wl in const track; do
  CMD="python $REPO/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu"
  rocprofv3 --pmc VALUBusy MemUnitStalled --kernel-trace -d $OUT/${wl}_busy -o bench -- $CMD > $OUT/${wl}_busy.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $OUT/${wl}_valu -o bench -- $CMD > $OUT/${wl}_valu.log 2>&1
  rocprofv3 --pmc LDSBankConflict SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/${wl}_lds -o bench -- $CMD > $OUT/${wl}_lds.log 2>&1
done
cd $REPO
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$OUT/*/*.db")):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(db, e); continue
    for r in rows:
        if "rows_kernel" in r[0] or "walk_kernel" in r[0]:
            print(db.split("/")[-2], r[0][:34], r[1], r[2], round(r[3], 3))
PY
