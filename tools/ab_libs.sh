#!/bin/bash
# Same-box A/B of several BUILDS of the library:  tools/ab_libs.sh ROUNDS "LIB1.so LIB2.so ..." "label|CASE ..." ["label|..." ...]
# ("cur" = doppler_amd/lib/libdoppler_hip.so as shipped is always among them), one process of tools/ab.py per build and round,
# round-robin, so that clock and thermal drift hit every build alike.
ROUNDS=$1; LIBS=$2; shift 2
cp doppler_amd/lib/libdoppler_hip.so /tmp/ab_cur.so
for r in $(seq $ROUNDS); do
  for v in cur $LIBS; do
    if [ $v = cur ]; then cp /tmp/ab_cur.so doppler_amd/lib/libdoppler_hip.so; else cp $v doppler_amd/lib/libdoppler_hip.so; fi
    echo "== $v, round $r"
    python tools/ab.py --rounds 5 "$@" 2>/dev/null | sed 's/^/  /'
  done
done
cp /tmp/ab_cur.so doppler_amd/lib/libdoppler_hip.so
