#!/bin/bash
# Counters behind the per-format-pair workgroup shapes of the span kernel (and the rows kernel / the replay for scale).
# One rocprofv3 --pmc pass per counter set and case (kernel trace only), tools/prof_case.py as the workload.
#   tools/prof_pairs.sh [ITERS]   -> gpurun_out/r04_pairs_pmc/{raw.tsv,summary.md}
set -u
ITERS=${1:-40}
REPO=$PWD
OUT=$REPO/gpurun_out/r04_pairs_pmc
mkdir -p $OUT
export TMPDIR=/tmp
SETS=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR"
 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
 "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE TCC_BUSY_sum"
)
: > $OUT/raw.tsv
run_case() {   # label case [opts...]
  local label=$1; shift
  local i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rm -rf /tmp/pp; mkdir -p /tmp/pp; cd /tmp
    rocprofv3 --pmc $set --kernel-trace -d /tmp/pp -o run -- python $REPO/tools/prof_case.py "$@" iters=$ITERS > /tmp/pp/log 2>&1
    cd $REPO
    python - "$label" $i <<'PY' >> $OUT/raw.tsv
import glob, sqlite3, sys
label, p = sys.argv[1], sys.argv[2]
dbs = glob.glob("/tmp/pp/**/*.db", recursive=True)
if not dbs:
    print("%s\tpass%s\tERROR\tno database\t0\t0" % (label, p)); sys.exit()
c = sqlite3.connect(dbs[0])
try:
    kern = c.execute("select name, count(*), avg(duration)/1e3 from kernels where name like '%dpx::%' and name not like '%build_lut%' group by name order by sum(duration) desc limit 1").fetchone()
    rows = c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name = ? group by counter_name", (kern[0],)).fetchall()
except Exception as e:
    print("%s\tpass%s\tERROR\t%s\t0\t0" % (label, p, str(e).replace("\t", " "))); sys.exit()
print("%s\tpass%s\tduration_us\t%s\t%d\t%.2f" % (label, p, kern[0][:60], kern[1], kern[2]))
for name, n, v in rows:
    print("%s\tpass%s\t%s\t%s\t%d\t%.1f" % (label, p, name, kern[0][:60], n, v))
PY
  done
}
run_case "rows 5000 Hz i16->i16 (headline)" const5000
run_case "span 5001 Hz i16->i16 plan=launch (4 waves x spans of 8)" const5001
run_case "span 5001 Hz i16->i16 spans of 4, 4 waves" const5001 walk_span=4 walk_waves=4
for pair in f32:f32 i16:f32 f32:i16; do
  run_case "span 5001 Hz $pair plan shape (4 waves x spans of 8)" const5001 pair=$pair walk_span=8 walk_waves=4
  run_case "span 5001 Hz $pair launch cut (default)" const5001 pair=$pair
done
run_case "span 5001 Hz i16:f32 spans of 4, 4 waves" const5001 pair=i16:f32 walk_span=4 walk_waves=4
run_case "span 5001 Hz i16:f32 spans of 8, 2 waves" const5001 pair=i16:f32 walk_span=8 walk_waves=2
run_case "span replay 600 s i16->i16" track600
run_case "span replay 300 s i16->f32" track300f pair=i16:f32
python tools/summarize_pairs.py $OUT/raw.tsv > $OUT/summary.md
cat $OUT/summary.md
