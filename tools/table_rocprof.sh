#!/bin/bash
# One rocprofv3 --kernel-trace --stats run per row of DESIGN.md's kernel table (tools/prof_case.py, ITERS launches each,
# back to back); prints a markdown table of the kernels' median / average / best duration per launch and the algorithmic
# GB/s they mean.  The first ~50-100 ms after idle run at ramping clocks (profiles/r02_walk.md), which is what separates
# the average from the median on the arithmetic-heavy rows.  A plan whose span launch is cut into sub-launches (long streams)
# counts as ONE launch per run: its sub-launches' durations are added up.
set -u
REPO=$PWD
export TMPDIR=/tmp
OUTMD=$REPO/gpurun_out/${ROUND:-r04}_table_rocprof${SUFFIX:-}.md
ITERS=${ITERS:-300}
echo "| row | kernel | launches | median us | avg us | min us | GB/s (median) | % of 8 TB/s (median) | % (avg) | % (best) |" > $OUTMD
echo "|---|---|---|---|---|---|---|---|---|---|" >> $OUTMD
# ONLY='regex' keeps the rows whose label matches (a quick subset between full tables)
row() {   # label bytes_per_sample case [opts...]
  local label=$1; local bps=$2; shift 2
  if [ -n "${ONLY:-}" ] && ! [[ "$label" =~ $ONLY ]]; then return; fi
  rm -rf /tmp/tr; mkdir -p /tmp/tr; cd /tmp
  rocprofv3 --kernel-trace --stats -d /tmp/tr -o run -- python $REPO/tools/prof_case.py "$@" iters=$ITERS > /tmp/tr/log 2>&1
  cd $REPO
  python - "$label" $bps $ITERS <<'PY' >> $OUTMD
import glob, sqlite3, sys, re
label, bps, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
db = glob.glob("/tmp/tr/**/*.db", recursive=True)
m = re.search(r"^ran .* (\d+)$", open("/tmp/tr/log").read(), re.M)
n = int(m.group(1)) if m else 0
if not db or not n:
    print("| %s | (no database / run failed) | | | | | | |" % label); sys.exit()
c = sqlite3.connect(db[0])
names = [r[0] for r in c.execute("select name, count(*) from kernels where name like '%dpx::%' group by name having count(*) >= 10 order by sum(duration) desc")]
per = {}
for nm in names:
    d = [r[0] / 1e3 for r in c.execute("select duration from kernels where name = ? order by start", (nm,))]
    g = len(d) // iters if iters and len(d) % iters == 0 else 1     # a long span launch is dealt out as g sub-launches per run of the plan
    per[nm] = [sum(d[i * g:(i + 1) * g]) for i in range(len(d) // g)]
k = min(len(v) for v in per.values())
tot = [sum(per[nm][i] for nm in names) for i in range(k)]           # a plan may be two launches: add them per run
tot_sorted = sorted(tot)
med, avg, best = tot_sorted[k // 2], sum(tot) / k, tot_sorted[0]
name = re.sub(r"\(.*", "", names[0]).replace("void ", "")
f = lambda us: n * bps / us / 1e3 / 80
print("| %s | `%s`%s | %d | %.1f | %.1f | %.1f | %.0f | %.1f | %.1f | %.1f |" % (label, name[:60], " (+%d more)" % (len(names) - 1) if len(names) > 1 else "",
                                                                 k, med, avg, best, n * bps / med / 1e3, f(med), f(avg), f(best)))
PY
}
if [ "${ROWSET:-default}" = generality ]; then
# round 5: the workloads the replay path was NOT tuned on — the reference README's own sample rates
# (256 / 300 ksps, /root/reference/README.md:53-62), 2.4 Msps, configs[4]'s real per-rank chunks, and the stream-size sweep
for rate in 256000 300000 1024000 2400000; do
  row "track replay 600 s at $rate sps, i16->i16" 8 track600 rate=$rate
  row "track replay 600 s at $rate sps, f32->i16" 12 track600 rate=$rate pair=f32:i16
done
row "per sample: track replay 600 s at 256000 sps, i16->i16" 8 track600 rate=256000 variant=1
row "configs[4] rank 0 chunk (450 s of 1 h, f32->i16)" 12 config4r0
row "configs[4] rank 3 chunk (450 s of 1 h, f32->i16)" 12 config4r3
row "configs[4] rank 3 chunk, span kernel forced (variant 5)" 12 config4r3 variant=5
for lg in 28 29 30 31; do
  row "span: 5001 Hz, 2^$lg samples, i16->i16" 8 const5001 n=$((1 << lg))
done
for lg in 28 30 31; do
  row "rows: 5000 Hz, 2^$lg samples, i16->i16" 8 const5000 n=$((1 << lg))
done
row "span: 5001 Hz, 2^30 samples, f32->i16" 12 const5001 n=$((1 << 30)) pair=f32:i16
else
row "5000 Hz (headline), i16->i16" 8 const5000
row "5000 Hz, f32->f32" 16 const5000 pair=f32:f32
row "5000 Hz, i16->f32" 12 const5000 pair=i16:f32
row "5000 Hz, f32->i16" 12 const5000 pair=f32:i16
row "815 kHz at 2.4 Msps (P=480)" 8 const815000 rate=2400000
row "100 Hz (P=10240)" 8 const100
row "9876.543 Hz (P=2592)" 8 const9876.543
row "5001 Hz (P=113027, odd)" 8 const5001
row "5001 Hz, f32->f32" 16 const5001 pair=f32:f32
row "1234 Hz (P=107047)" 8 const1234
row "12345 Hz (P=40313)" 8 const12345
row "9999 Hz (P=177989)" 8 const9999
row "777 Hz (P=333426)" 8 const777
row "7777.77 Hz (P=28043)" 8 const7777.77
row "-5234.17 Hz (P=107405)" 8 const-5234.17
row "3 Hz (P=1024000)" 8 const3
row "3 Hz, sincos per sample (variant 1), i16->i16" 8 const3 variant=1
row "3 Hz, sincos per sample, f32->f32" 16 const3 variant=1 pair=f32:f32
row "5001 Hz, sincos per sample, i16->i16" 8 const5001 variant=1
row "5001 Hz, sincos per sample, f32->f32" 16 const5001 variant=1 pair=f32:f32
row "5001 Hz, sincos per sample, i16->f32" 12 const5001 variant=1 pair=i16:f32
row "5001 Hz, sincos per sample, f32->i16" 12 const5001 variant=1 pair=f32:i16
row "3 Hz, sincos per sample, i16->f32" 12 const3 variant=1 pair=i16:f32
row "3 Hz, sincos per sample, f32->i16" 12 const3 variant=1 pair=f32:i16
row "track replay 600 s, i16->i16" 8 track600
row "legacy i16 cast: 5000 Hz (rows kernel), i16->i16" 8 const5000 cast=legacy
row "legacy i16 cast: 5001 Hz (span kernel), i16->i16" 8 const5001 cast=legacy
row "legacy i16 cast: track replay 600 s, i16->i16" 8 track600 cast=legacy
row "track replay 300 s, f32->i16" 12 track300f pair=f32:i16
row "track replay 300 s, f32->f32" 16 track300f pair=f32:f32
row "track replay 300 s, i16->f32" 12 track300f pair=i16:f32
# the same replays and the odd-period const stream with every corrector evaluated per sample (tile kernel only)
row "per sample: track replay 300 s, f32->i16" 12 track300f pair=f32:i16 variant=1
row "per sample: track replay 300 s, f32->f32" 16 track300f pair=f32:f32 variant=1
row "per sample: track replay 300 s, i16->f32" 12 track300f pair=i16:f32 variant=1
row "per sample: track replay 600 s, i16->i16" 8 track600 variant=1
row "span kernel forced (variant 5): track replay 300 s, f32->i16" 12 track300f pair=f32:i16 variant=5
row "span: 5001 Hz, f32->i16" 12 const5001 pair=f32:i16
row "span: 5001 Hz, i16->f32" 12 const5001 pair=i16:f32
fi
cat $OUTMD
