#!/bin/bash
# Same-box A/B of several builds on the replay's row classes:  tools/ab_classes.sh ROUNDS LIB1.so LIB2.so ...   ("cur" is always included)
ROUNDS=$1; shift
cp doppler_amd/lib/libdoppler_hip.so /tmp/abc_cur.so
for r in $(seq $ROUNDS); do
  for v in cur "$@"; do
    if [ $v = cur ]; then cp /tmp/abc_cur.so doppler_amd/lib/libdoppler_hip.so; else cp $v doppler_amd/lib/libdoppler_hip.so; fi
    echo "== $v $r"
    OPTS="${OPTS:-[{\}]}" python tools/replay_classes.py 2>/dev/null
  done
done
cp /tmp/abc_cur.so doppler_amd/lib/libdoppler_hip.so
