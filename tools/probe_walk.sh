for sh in 5000 815000 9876.543 5001 3; do for pair in i16:i16 f32:f32; do echo -n "shift $sh $pair: "; python tools/sweep.py --shift=$sh --pairs $pair --variants 3 --iters 40 $( [ $sh = 815000 ] && echo --rate 2400000 ) 2>&1 | grep '"shift"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['GBps_avg']), 'best', round(d['GBps_best']))"; done; done
for i in 1 2 3; do python bench.py --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['achieved'])"; done
