#!/bin/bash
# Same-box A/B of several BUILDS of the library:  tools/ab_libs.sh ROUNDS "LIB1.so LIB2.so ..." "label|CASE ..." ["label|..." ...]
# ("cur" = doppler_amd/lib/libdoppler_hip.so as shipped is always among them), one process of tools/ab.py per build and round,
# round-robin, so that clock and thermal drift hit every build alike.  The alternates live OUTSIDE the tree (build them on the
# GPU box: `git stash; make lib; cp doppler_amd/lib/libdoppler_hip.so /tmp/before.so; git stash pop; make lib`): doppler_amd/lib
# holds the product library and nothing else (tests/test_host_logic.py::test_only_the_product_library_ships), and this script
# puts the shipped one back when it is done, or when it is interrupted.
ROUNDS=$1; LIBS=$2; shift 2
cp doppler_amd/lib/libdoppler_hip.so /tmp/ab_cur.so
trap 'cp /tmp/ab_cur.so doppler_amd/lib/libdoppler_hip.so' EXIT
for v in $LIBS; do case $v in /tmp/*) ;; *) echo "alternate $v must live under /tmp" >&2; exit 2;; esac; done
for r in $(seq $ROUNDS); do
  for v in cur $LIBS; do
    if [ $v = cur ]; then cp /tmp/ab_cur.so doppler_amd/lib/libdoppler_hip.so; else cp $v doppler_amd/lib/libdoppler_hip.so; fi
    echo "== $v, round $r"
    python tools/ab.py --rounds 5 "$@" 2>/dev/null | sed 's/^/  /'
  done
done
cp /tmp/ab_cur.so doppler_amd/lib/libdoppler_hip.so
