set -u
EXE=doppler_amd/bin/doppler
F=/dev/shm/dpx_in.iq
python - <<PY
import numpy as np
rng = np.random.default_rng(1)
rng.integers(-23170, 23171, size=1 << 31, dtype=np.int16).tofile("$F")     # 4 GiB
PY
run() {
  local label=$1; local out=$2; shift; shift
  s=$(date +%s.%N)
  env DOPPLER_STATS=1 "$@" $EXE const -s 1024000 -i i16 --shift 5000 < $F 2>/tmp/dpx_err > $out
  e=$(date +%s.%N)
  st=$(grep "doppler stats: [0-9]" /tmp/dpx_err | sed 's/.*= \([0-9.]*\) Msamples.*/\1/')
  how=$(grep "doppler stats: [0-9]" /tmp/dpx_err | sed 's/.*slabs of/slabs of/')
  python -c "t=$e-$s; print('%-50s steady %8s Msamples/s   whole %.3f s  %s' % ('$label', '$st', t, '''$how'''))"
}
cat /proc/loadavg
run "warm-up" /dev/shm/dpx_out.iq
for rep in 1 2; do
run "file -> /dev/null, defaults" /dev/null
run "file -> /dev/null, 16 threads 32M" /dev/null DOPPLER_IO_THREADS=16 DOPPLER_SLAB_BYTES=33554432
run "file -> /dev/null, 32 threads 32M" /dev/null DOPPLER_IO_THREADS=32 DOPPLER_SLAB_BYTES=33554432
run "file -> existing tmpfs file, defaults" /dev/shm/dpx_out.iq
run "file -> existing tmpfs file, 16 thr 32M" /dev/shm/dpx_out.iq DOPPLER_IO_THREADS=16 DOPPLER_SLAB_BYTES=33554432
rm -f /dev/shm/dpx_out2.iq
run "file -> fresh tmpfs file, 16 thr 32M" /dev/shm/dpx_out2.iq DOPPLER_IO_THREADS=16 DOPPLER_SLAB_BYTES=33554432
run "file -> fresh tmpfs file, 16 thr 32M, pwrite" /dev/shm/dpx_out2.iq DOPPLER_IO_THREADS=16 DOPPLER_SLAB_BYTES=33554432 DOPPLER_NO_MMAP=1
done
rm -f $F /dev/shm/dpx_out.iq /dev/shm/dpx_out2.iq
