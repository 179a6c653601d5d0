#!/bin/bash
# SQ / TCC counters of one tools/prof_case.py invocation, separate --pmc passes (kernel trace only).
#   [QUICK=1] tools/prof_sq.sh TAG CASE [key=value ...]      -> gpurun_out/sq_TAG/summary.txt (QUICK: first pass only)
set -u
TAG=$1; shift
REPO=$PWD
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/tools/prof_case.py $*"
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE")
if [ -z "${QUICK:-}" ]; then
  SETS+=("SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o run -- $CMD > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<PY > $OUT/summary.txt
import sqlite3, glob
for db in sorted(glob.glob("$OUT/*/*.db")):
    c = sqlite3.connect(db)
    try:
        for r in c.execute("select name, count(*), avg(duration)/1e3, min(duration)/1e3 from kernels group by name order by sum(duration) desc limit 2"):
            print("$TAG", db.split("/")[-2], "duration_us", r[0][:40], r[1], round(r[2], 1), round(r[3], 1))
    except Exception as e:
        print("no kernels view", e)
    try:
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(db, e); continue
    for r in rows:
        if "rows_kernel" in r[0] or "walk_kernel" in r[0] or "tile_kernel" in r[0]:
            print("$TAG", r[0][:40], r[1], r[2], round(r[3], 1))
PY
cat $OUT/summary.txt
find $OUT -name "*.db" -delete
