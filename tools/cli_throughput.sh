#!/bin/bash
# End-to-end throughput of the `doppler` command on the GPU box (PCIe- and pipe-inclusive; never the bench value).
# Input lives in /dev/shm so that storage is not what is measured.
set -u
REPO=$PWD
EXE=$REPO/doppler_amd/bin/doppler
F=/dev/shm/dpx_in.iq
python - <<PY
import numpy as np
rng = np.random.default_rng(1)
a = rng.integers(-23170, 23171, size=1 << 29, dtype=np.int16)     # 1 GiB of i16 IQ = 268435456 samples
a.tofile("$F")
PY
ls -la $F
for slab in 4194304 33554432 134217728; do
  for mode in file pipe; do
    s=$(date +%s.%N)
    if [ $mode = file ]; then DOPPLER_SLAB_BYTES=$slab $EXE const -s 1024000 -i i16 --shift 5000 < $F > /dev/shm/dpx_out.iq 2>/dev/null
    else cat $F | DOPPLER_SLAB_BYTES=$slab $EXE const -s 1024000 -i i16 --shift 5000 2>/dev/null | cat > /dev/null; fi
    e=$(date +%s.%N)
    python -c "t=$e-$s; print('slab %9d %-4s  %.3f s  %.0f Msamples/s  %.2f GB/s in+out' % ($slab, '$mode', t, 268.435456/t, 2*1.073741824/t))"
  done
done
rm -f $F /dev/shm/dpx_out.iq
