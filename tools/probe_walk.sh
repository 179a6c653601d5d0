for v in 3; do python tools/sweep.py --track 300 --variants $v --pairs i16:i16,f32:i16,i16:f32,f32:f32 2>&1 | grep kernel | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('variant', d['variant'], d['pair'], 'ms', d['ms_avg'], 'GB/s', d['GBps_avg'])"; done
