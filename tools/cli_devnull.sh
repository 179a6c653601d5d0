#!/bin/bash
# doppler file -> /dev/null for several worker counts / slab sizes (the read + PCIe + kernel side of the command)
REPO=$PWD; EXE=$REPO/doppler_amd/bin/doppler; F=/dev/shm/dpx_in.iq
python - <<PY
import numpy as np
np.random.default_rng(1).integers(-23170, 23171, size=1 << 31, dtype=np.int16).tofile("$F")
PY
for thr in 4 8 12 16 24; do for slab in 8388608 16777216 33554432; do
  echo -n "io_threads=$thr slab=$slab: "; DOPPLER_STATS=1 DOPPLER_IO_THREADS=$thr DOPPLER_SLAB_BYTES=$slab $EXE const -s 1024000 -i i16 --shift 5000 < $F 2>&1 >/dev/null | grep stats | sed 's/.*= \([0-9.]*\) Msamples.*/\1 Msamples\/s/'
done; done
echo -n "2 contexts, io_threads=16 slab=16M: "; DOPPLER_DEVICES=0,0 DOPPLER_STATS=1 DOPPLER_IO_THREADS=16 DOPPLER_SLAB_BYTES=16777216 $EXE const -s 1024000 -i i16 --shift 5000 < $F 2>&1 >/dev/null | grep stats | sed 's/.*= \([0-9.]*\) Msamples.*/\1 Msamples\/s/'
rm -f $F
