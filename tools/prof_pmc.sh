#!/bin/bash
# rocprofv3 --pmc passes (kernel trace only: one pass per counter set, never together with a runtime trace) over cases of
# tools/prof_case.py; per case and counter the per-launch average of the dominant dpx kernel.
#   tools/prof_pmc.sh NAME "label|case [opts...]" ...        -> gpurun_out/NAME/raw.tsv (+ table.md: one row per case)
#   SETS="memory" (default: requests, stalls, queue levels, address translation) | "all" (adds the SQ sets of round 4)
set -u
NAME=$1; shift
ITERS=${ITERS:-30}
REPO=$PWD
OUT=$REPO/gpurun_out/$NAME
mkdir -p $OUT
export TMPDIR=/tmp
SETS=(
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
 "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE TCC_BUSY_sum"
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_UTCL2_BUSY"
)
if [ "${SETS_EXTRA:-}" = all ]; then
 SETS+=("SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM")
fi
: > $OUT/raw.tsv
for spec in "$@"; do
  label=${spec%%|*}; args=${spec#*|}
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rm -rf /tmp/pp; mkdir -p /tmp/pp; cd /tmp
    rocprofv3 --pmc $set --kernel-trace -d /tmp/pp -o run -- python $REPO/tools/prof_case.py $args iters=$ITERS > /tmp/pp/log 2>&1
    cd $REPO
    python - "$label" $i <<'PY' >> $OUT/raw.tsv
import glob, re, sqlite3, sys
label, p = sys.argv[1], sys.argv[2]
dbs = glob.glob("/tmp/pp/**/*.db", recursive=True)
m = re.search(r"^ran .* (\d+)$", open("/tmp/pp/log").read(), re.M)
if not dbs or not m:
    print("%s\tpass%s\tERROR\tno database / run failed\t0\t0" % (label, p)); sys.exit()
c = sqlite3.connect(dbs[0])
try:
    kern = c.execute("select name, count(*), avg(duration)/1e3 from kernels where name like '%dpx::%' and name not like '%build_lut%' and name not like '%copy_kernel%' group by name order by sum(duration) desc limit 1").fetchone()
    rows = c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name = ? group by counter_name", (kern[0],)).fetchall()
except Exception as e:
    print("%s\tpass%s\tERROR\t%s\t0\t0" % (label, p, str(e).replace("\t", " "))); sys.exit()
print("%s\tpass%s\tn_samples\t%s\t1\t%s" % (label, p, kern[0][:60], m.group(1)))
print("%s\tpass%s\tduration_us\t%s\t%d\t%.2f" % (label, p, kern[0][:60], kern[1], kern[2]))
for name, n, v in rows:
    print("%s\tpass%s\t%s\t%s\t%d\t%.1f" % (label, p, name, kern[0][:60], n, v))
PY
  done
done
python - $OUT/raw.tsv > $OUT/table.md <<'PY'
import collections, sys
rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    p = line.rstrip("\n").split("\t")
    if len(p) < 6: continue
    label, pas, name, kern, n, v = p
    d = rows.setdefault(label, {"kernel": kern, "dur": []})
    if name == "duration_us": d["dur"].append(float(v))
    elif name == "ERROR": d.setdefault("err", []).append(pas + " " + kern)
    else: d[name] = float(v)
cols = ["case", "kernel", "us (fastest pass)", "EA rd req /Ksample", "EA wr req /Ksample", "rd DRAM-credit stall /req", "wr DRAM-credit stall /req", "rd req in flight", "wr req in flight",
        "UTCL1 req /Ksample", "UTCL1 miss /Ksample", "miss under miss /Ksample", "UTCL2-credit stall cyc /Ksample", "inflight-max stall cyc /Ksample", "GRBM_UTCL2_BUSY / GUI_ACTIVE", "VALU instr /sample", "WAIT_ANY % of wave cycles"]
print("| " + " | ".join(cols) + " |"); print("|" + "---|" * len(cols))
for label, d in rows.items():
    g = lambda k: d.get(k, float("nan"))
    ks = g("n_samples") / 1000.0
    gui = g("GRBM_GUI_ACTIVE")
    print("| " + " | ".join([label, "`%s`" % d["kernel"][:40], "%.1f" % (min(d["dur"]) if d["dur"] else float("nan")),
        "%.2f" % (g("TCC_EA0_RDREQ_sum") / ks), "%.2f" % (g("TCC_EA0_WRREQ_sum") / ks),
        "%.3f" % (g("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum") / g("TCC_EA0_RDREQ_sum")), "%.3f" % (g("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum") / g("TCC_EA0_WRREQ_sum")),
        "%.0f" % (g("TCC_EA0_RDREQ_LEVEL_sum") / gui), "%.0f" % (g("TCC_EA0_WRREQ_LEVEL_sum") / gui),
        "%.2f" % (g("TCP_UTCL1_REQUEST_sum") / ks), "%.3f" % (g("TCP_UTCL1_TRANSLATION_MISS_sum") / ks), "%.3f" % (g("TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum") / ks),
        "%.2f" % (g("TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum") / ks), "%.2f" % (g("TCP_UTCL1_STALL_INFLIGHT_MAX_sum") / ks), "%.3f" % (g("GRBM_UTCL2_BUSY") / gui),
        "%.1f" % (g("SQ_INSTS_VALU") * 64 / g("n_samples")), "%.0f" % (100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"))]) + " |")
    for e in d.get("err", []): print("| %s: %s |" % (label, e))
PY
cat $OUT/table.md
