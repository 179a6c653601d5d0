#!/bin/bash
# Same-box A/B of several builds of the library:  tools/ab_libs.sh "ab.py arguments" ROUNDS LIB1.so LIB2.so ...
# ("cur" = doppler_amd/lib/libdoppler_hip.so as shipped), one process per run, round-robin.
ARGS=$1; ROUNDS=$2; shift 2
cp doppler_amd/lib/libdoppler_hip.so /tmp/ab_cur.so
for r in $(seq $ROUNDS); do
  for v in cur "$@"; do
    if [ $v = cur ]; then cp /tmp/ab_cur.so doppler_amd/lib/libdoppler_hip.so; else cp $v doppler_amd/lib/libdoppler_hip.so; fi
    echo "== $v $r"
    python tools/ab.py $ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  %-40s %-8s %-9s %-36s %6.1f' % (d['case'][:40], d['pair'], 'x'.join(map(str, d.get('geom', []))), str(d.get('opts'))[:36], d['pct_peak']))"
  done
done
cp /tmp/ab_cur.so doppler_amd/lib/libdoppler_hip.so
