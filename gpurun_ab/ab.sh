for rep in 1 2 3; do for v in 4,2 5,2; do DPX_WALK_SHAPE=$v python tools/sweep.py --track 300 --variants 3 --pairs i16:i16,f32:i16,i16:f32,f32:f32 2>&1 | grep kernel | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['pair'], d['GBps_avg'])"; done; done | sort | awk '{k=$1" "$2; s[k]+=$3; n[k]++} END{for(k in s) print k, s[k]/n[k]}' | sort
for v in 4,2 5,2; do echo "== $v"; DPX_WALK_SHAPE=$v python tools/track_probe.py 2>&1 | grep auto | cut -c1-40,95-140; done
