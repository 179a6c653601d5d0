#!/bin/bash
# Where the end-to-end wall is: raw tmpfs copy, doppler to /dev/null, to a preallocated file, to a fresh file.
REPO=$PWD; EXE=$REPO/doppler_amd/bin/doppler; F=/dev/shm/dpx_in.iq; O=/dev/shm/dpx_out.iq
python - <<PY
import numpy as np
np.random.default_rng(1).integers(-23170, 23171, size=1 << 30, dtype=np.int16).tofile("$F")
PY
t() { s=$(date +%s.%N); "$@"; e=$(date +%s.%N); python -c "print('   %.3f s -> %.2f GB/s per direction, %.0f Msamples/s' % ($e-$s, 2.147483648/($e-$s), 536.870912/($e-$s)))"; }
echo "dd tmpfs -> fresh tmpfs file (bs=8M)"; rm -f $O; t dd if=$F of=$O bs=8M status=none
echo "dd tmpfs -> existing tmpfs file (conv=notrunc)"; t dd if=$F of=$O bs=8M conv=notrunc status=none
echo "dd tmpfs -> /dev/null"; t dd if=$F of=/dev/null bs=8M status=none
for thr in 1 4 8; do
  echo "doppler file -> /dev/null, io_threads=$thr"; DOPPLER_STATS=1 DOPPLER_IO_THREADS=$thr $EXE const -s 1024000 -i i16 --shift 5000 < $F 2>&1 >/dev/null | grep stats
  echo "doppler file -> existing file (no truncation), io_threads=$thr"; DOPPLER_STATS=1 DOPPLER_IO_THREADS=$thr $EXE const -s 1024000 -i i16 --shift 5000 < $F 2>&1 1<>$O | grep stats
  rm -f $O
  echo "doppler file -> fresh file, io_threads=$thr"; DOPPLER_STATS=1 DOPPLER_IO_THREADS=$thr $EXE const -s 1024000 -i i16 --shift 5000 < $F 2>&1 >$O | grep stats
done
rm -f $F $O
