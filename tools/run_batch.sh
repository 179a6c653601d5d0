set -u
mkdir -p gpurun_out
timeout 1200 python tools/ab.py --set pairshape --rounds 10 --iters 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-24s %-8s %-40s %-40s %6.1f' % (d['case'][:24], d['pair'], str(d['opts']), d.get('kernel', '')[:40], d['pct_peak']))" | tee gpurun_out/r04b_pairshape.log
