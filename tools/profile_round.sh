#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats and the two HBM-traffic PMC passes for the bench
# command (headline and track workloads), plus a calibration pass on the plain copy kernel; the summaries
# (gpurun_out/<round>_*.md / .json, small) are what comes back — copy them into profiles/.
#   tools/profile_round.sh r05
# The headline's kernel trace is the driver's own command shape (--steps 20 --warmup 5, sustained leg included): the summary quotes
# the first 1 + W + K launches (one-shot, warm-up, timed region: what `roofline.frac` times) and the sustained launches apart.
# The track workload is profiled over 300 launches (bench.py's own default for `--workload track`): the clocks need
# 50-150 ms under load to settle, and the summary quotes the launches after the first 160 ms ("settled") beside the
# average over all of them, so that it can be compared with the settled figure `extra.track` carries in the bench line.
set -u
R=${1:-r05}
REPO=$PWD
OUT=/tmp/prof_$R
rm -rf $OUT; mkdir -p $OUT $REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
for wl in const track track_256k config4_chunk; do
  STEPS=20; [ $wl != const ] && STEPS=300
  CMD="python $REPO/bench.py --workload $wl --steps $STEPS --warmup 5 --no-cpu --no-extra"
  rocprofv3 --kernel-trace --stats -d $OUT/${wl}_trace -o bench -- $CMD > $OUT/${wl}_trace.log 2>&1
  CMD="python $REPO/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-extra --no-sustain"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${wl}_fetch -o bench -- $CMD > $OUT/${wl}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${wl}_write -o bench -- $CMD > $OUT/${wl}_write.log 2>&1
done
# calibration: a copy of known size through dpx_debug_copy (1 GiB read, 1 GiB written)
CAL="python $REPO/tools/prof_case.py copy iters=5"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cal_fetch -o cal -- $CAL > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cal_write -o cal -- $CAL > $OUT/cal_write.log 2>&1
cd $REPO
grep -h "^{" $OUT/const_trace.log | tail -1 > gpurun_out/${R}_bench_line_under_rocprof.json
grep -h "^{" $OUT/track_trace.log | tail -1 > gpurun_out/${R}_bench_line_track_under_rocprof.json
python tools/summarize_round.py $OUT gpurun_out $R
