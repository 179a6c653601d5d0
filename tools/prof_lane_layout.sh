#!/bin/bash
# Counters behind the tile kernel's lane layouts (round 4): the per-sample path of the pairs with an f32 side, the library
# before the change (four consecutive samples per lane: 32 bytes per lane on an f32 side) against the shipped one (two pairs
# half a block apart: every 16-byte vector of a wavefront instruction adjacent to its neighbours').
# Only counter sets that prof_pairs.sh has run before: a pass with TCC_REQ / TCC_HIT / TCC_MISS did not come back in 30 minutes.
#   tools/prof_lane_layout.sh OLD_LIB.so [ITERS]   -> gpurun_out/r04_lane_layout_pmc.tsv
set -u
OLD=$1; ITERS=${2:-40}
REPO=$PWD
OUT=$REPO/gpurun_out/r04_lane_layout_pmc.tsv
export TMPDIR=/tmp
SETS=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
 "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE TCC_BUSY_sum"
)
cp doppler_amd/lib/libdoppler_hip.so /tmp/cur.so
: > $OUT
for lib in old new; do
  if [ $lib = old ]; then cp $OLD doppler_amd/lib/libdoppler_hip.so; else cp /tmp/cur.so doppler_amd/lib/libdoppler_hip.so; fi
  for pair in i16:f32 f32:f32 f32:i16; do
    i=0
    for set in "${SETS[@]}"; do
      i=$((i+1))
      rm -rf /tmp/pp; mkdir -p /tmp/pp; cd /tmp
      timeout 180 rocprofv3 --pmc $set --kernel-trace -d /tmp/pp -o run -- python $REPO/tools/prof_case.py const3 variant=1 pair=$pair iters=$ITERS > /tmp/pp/log 2>&1
      cd $REPO
      python - "$lib $pair" $i <<'PY' >> $OUT
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
  done
done
cp /tmp/cur.so doppler_amd/lib/libdoppler_hip.so
cat $OUT
