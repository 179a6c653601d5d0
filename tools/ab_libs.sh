#!/bin/bash
# Same-box A/B of two builds of the library:  tools/ab_libs.sh OTHER.so "ab.py arguments" [rounds]
# alternates doppler_amd/lib/libdoppler_hip.so (as shipped: "new") with OTHER.so ("old"), one process per run.
OTHER=$1; ARGS=$2; ROUNDS=${3:-2}
cp doppler_amd/lib/libdoppler_hip.so /tmp/ab_new.so
for r in $(seq $ROUNDS); do
  for v in new old; do
    if [ $v = old ]; then cp $OTHER doppler_amd/lib/libdoppler_hip.so; else cp /tmp/ab_new.so doppler_amd/lib/libdoppler_hip.so; fi
    echo "== $v $r"
    python tools/ab.py $ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  %-45s %-8s %-14s %6.1f' % (d['case'][:45], d['pair'], str(d.get('geom')), d['pct_peak']))"
  done
done
cp /tmp/ab_new.so doppler_amd/lib/libdoppler_hip.so
