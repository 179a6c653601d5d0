#!/bin/bash
# Why do the span and tile kernels lose 4-5 points right after a process starts (bench.py: roofline.frac against frac_settled)?
# Per launch of ONE fresh process: duration (kernel trace) and counters (separate --pmc passes, kernel trace only), early
# launches (the first 25) against settled ones (those starting more than 160 ms after the first).
#   tools/cold_probe.sh NAME "label|case [opts...]" ...      -> gpurun_out/NAME/cold.md (+ raw per-launch tsv)
# GRBM_GUI_ACTIVE / duration is the shader clock the launch ran at; SQ_* per wave cycle say what the wavefronts waited for.
set -u
NAME=$1; shift
ITERS=${ITERS:-300}
REPO=$PWD
OUT=$REPO/gpurun_out/$NAME
mkdir -p $OUT
export TMPDIR=/tmp
SETS=(
 "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES"
 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
)
: > $OUT/cold_raw.tsv
for spec in "$@"; do
  label=${spec%%|*}; args=${spec#*|}
  rm -rf /tmp/cp; mkdir -p /tmp/cp; cd /tmp
  rocprofv3 --kernel-trace -d /tmp/cp/trace -o run -- python $REPO/tools/prof_case.py $args iters=$ITERS > /tmp/cp/trace.log 2>&1
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d /tmp/cp/pmc$i -o run -- python $REPO/tools/prof_case.py $args iters=$ITERS > /tmp/cp/pmc$i.log 2>&1
  done
  cd $REPO
  python - "$label" >> $OUT/cold_raw.tsv <<'PY'
import glob, sqlite3, sys
label = sys.argv[1]
def db(d):
    f = glob.glob("/tmp/cp/%s/**/*.db" % d, recursive=True)
    return sqlite3.connect(f[0]) if f else None
def dominant(c):
    return c.execute("select name from kernels where name like '%dpx::%' and name not like '%build_lut%' and name not like '%copy_kernel%' group by name order by sum(duration) desc limit 1").fetchone()[0]
c = db("trace")
k = dominant(c)
rows = c.execute("select start, duration from kernels where name = ? order by start", (k,)).fetchall()
t0 = rows[0][0]
for idx, (st, du) in enumerate(rows):
    print("%s\ttrace\t%d\t%.3f\tduration_us\t%.2f" % (label, idx, (st - t0) / 1e6, du / 1e3))
for p in range(1, 9):
    c = db("pmc%d" % p)
    if c is None: continue
    try:
        k = dominant(c)
        rows = c.execute("select dispatch_id, start, duration from kernels where name = ? order by start", (k,)).fetchall()
    except Exception as e:
        print("%s\tpmc%d\t0\t0\tERROR\t0" % (label, p)); continue
    t0 = rows[0][1]
    order = {r[0]: (i, (r[1] - t0) / 1e6, r[2] / 1e3) for i, r in enumerate(rows)}
    for did, (i, ms, du) in order.items():
        print("%s\tpmc%d\t%d\t%.3f\tduration_us\t%.2f" % (label, p, i, ms, du))
    try:
        cr = c.execute("select dispatch_id, counter_name, value from counters_collection where kernel_name = ?", (k,)).fetchall()
    except Exception as e:
        continue
    for did, name, v in cr:
        if did in order:
            print("%s\tpmc%d\t%d\t%.3f\t%s\t%.1f" % (label, p, order[did][0], order[did][1], name, v))
PY
done
python - $OUT/cold_raw.tsv > $OUT/cold.md <<'PY'
import collections, statistics, sys
data = collections.defaultdict(lambda: collections.defaultdict(dict))      # label -> (pass, idx) -> name -> value ; plus time
for line in open(sys.argv[1]):
    p = line.rstrip("\n").split("\t")
    if len(p) != 6 or p[4] == "ERROR": continue
    label, pas, idx, ms, name, v = p
    d = data[label][(pas, int(idx))]
    d[name] = float(v); d["_ms"] = float(ms)
def med(xs): return statistics.median(xs) if xs else float("nan")
print("| case | phase | launches | duration us (no counters) | duration us (under --pmc) | shader clock MHz (GUI_ACTIVE / duration) | GUI_ACTIVE cycles | SQ_BUSY / GUI_ACTIVE | wave cycles / launch (M) | WAIT_INST_ANY % of wave cycles | WAIT_ANY % | ACTIVE_INST_VALU % of wave cycles | ACTIVE_INST_VMEM % | VALU instr / launch (M) | EA rd requests in flight |")
print("|" + "---|" * 15)
for label, d in data.items():
    for phase, sel in (("first 25", lambda i, ms: i < 25), ("settled (> 160 ms)", lambda i, ms: ms > 160.0)):
        def col(pas, name):
            return [v[name] for (p, i), v in d.items() if p == pas and name in v and sel(i, v["_ms"])]
        tr = col("trace", "duration_us")
        pm = col("pmc1", "duration_us")
        gui, busy, wavec = col("pmc1", "GRBM_GUI_ACTIVE"), col("pmc1", "SQ_BUSY_CYCLES"), col("pmc1", "SQ_WAVE_CYCLES")
        clk = [g / t for g, t in zip(gui, pm)] if gui and len(gui) == len(pm) else []
        wia, aiv, niv, wa = col("pmc2", "SQ_WAIT_INST_ANY"), col("pmc2", "SQ_ACTIVE_INST_VALU"), col("pmc2", "SQ_INSTS_VALU"), col("pmc2", "SQ_WAIT_ANY")
        lvl, gui3 = col("pmc3", "TCC_EA0_RDREQ_LEVEL_sum"), col("pmc3", "GRBM_GUI_ACTIVE")
        avm = col("pmc4", "SQ_ACTIVE_INST_VMEM")
        wc = med(wavec)
        print("| %s | %s | %d | %.1f | %.1f | %.0f | %.0f | %.3f | %.1f | %.1f | %.1f | %.1f | %.1f | %.2f | %.0f |" % (
            label, phase, len(tr), med(tr), med(pm), med(clk), med(gui), med(busy) / med(gui) if gui else float("nan"), wc / 1e6,
            100 * med(wia) / wc, 100 * med(wa) / wc, 100 * med(aiv) / wc, 100 * med(avm) / wc, med(niv) / 1e6, med(lvl) / med(gui3) if gui3 else float("nan")))
PY
cat $OUT/cold.md
