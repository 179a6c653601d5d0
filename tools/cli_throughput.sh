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
for slab in 1048576 4194304 8388608 33554432; do
  for mode in file pipe; do
    s=$(date +%s.%N)
    if [ $mode = file ]; then DOPPLER_STATS=1 DOPPLER_SLAB_BYTES=$slab $EXE const -s 1024000 -i i16 --shift 5000 < $F 2>/tmp/dpx_err > /dev/shm/dpx_out.iq
    else cat $F | DOPPLER_STATS=1 DOPPLER_SLAB_BYTES=$slab $EXE const -s 1024000 -i i16 --shift 5000 2>/tmp/dpx_err | cat > /dev/null; fi
    grep "doppler stats" /tmp/dpx_err
    e=$(date +%s.%N)
    python -c "t=$e-$s; print('slab %9d %-4s  %.3f s  %.0f Msamples/s  %.2f GB/s in+out' % ($slab, '$mode', t, 268.435456/t, 2*1.073741824/t))"
  done
done
rm -f $F /dev/shm/dpx_out.iq
