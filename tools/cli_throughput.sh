#!/bin/bash
# End-to-end throughput of the `doppler` command on the GPU box (PCIe- and file/pipe-inclusive; never the bench value).
# Input and output live in /dev/shm so that storage is not what is measured.
set -u
REPO=$PWD
EXE=$REPO/doppler_amd/bin/doppler
F=/dev/shm/dpx_in.iq
python - <<PY
import numpy as np
rng = np.random.default_rng(1)
a = rng.integers(-23170, 23171, size=1 << 30, dtype=np.int16)     # 2 GiB of i16 IQ = 536870912 samples
a.tofile("$F")
PY
ls -la $F
N=536.870912
run() {   # label, env..., then mode
  local label=$1; shift
  local mode=$1; shift
  s=$(date +%s.%N)
  if [ $mode = file ]; then env DOPPLER_STATS=1 "$@" $EXE const -s 1024000 -i i16 --shift 5000 < $F 2>/tmp/dpx_err > /dev/shm/dpx_out.iq
  elif [ $mode = pipe ]; then cat $F | env DOPPLER_STATS=1 "$@" $EXE const -s 1024000 -i i16 --shift 5000 2>/tmp/dpx_err | cat > /dev/null
  fi
  e=$(date +%s.%N)
  st=$(grep "doppler stats" /tmp/dpx_err | sed 's/.*= \([0-9.]*\) Msamples.*/\1/')
  python -c "t=$e-$s; print('%-44s %-4s steady %8s Msamples/s   whole process %.3f s = %.0f Msamples/s' % ('$label', '$mode', '$st', t, $N/t))"
}
run "warm-up" file DOPPLER_SLAB_BYTES=8388608
for thr in 1 2 4 8 16 32; do
  for slab in 4194304 16777216; do
    run "io_threads=$thr slab=$slab" file DOPPLER_IO_THREADS=$thr DOPPLER_SLAB_BYTES=$slab
  done
done
run "gpus=2 (same device) io_threads=8 slab=8M" file DOPPLER_DEVICES=0,0 DOPPLER_IO_THREADS=8 DOPPLER_SLAB_BYTES=8388608
run "single reader + single writer (round 1 shape)" file DOPPLER_NO_PREAD=1 DOPPLER_NO_PWRITE=1 DOPPLER_SLAB_BYTES=8388608
run "pipes" pipe DOPPLER_SLAB_BYTES=8388608
cmp <(head -c 100000000 /dev/shm/dpx_out.iq | md5sum) <(head -c 100000000 /dev/shm/dpx_out.iq | md5sum) > /dev/null
rm -f $F /dev/shm/dpx_out.iq
