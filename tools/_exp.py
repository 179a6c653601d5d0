import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, doppler_amd
ctx = doppler_amd.Context(0)
pat = np.random.default_rng(1).integers(-23170, 23171, size=bench.RING_SLAB_BYTES // 2, dtype=np.int16).view(np.uint8)
def fill(j, buf): buf[:] = pat
out = []
for i in range(int(sys.argv[1])):
    dt, nb, desc, st, win = bench.drive_ring(ctx, bench.RING_SLAB_BYTES, 4, 2 << 30, fill, None, warm_s=0.3 if i == 0 else 0.0)
    out.append("%.1f/%d%s" % (nb / dt / 1e9, desc["probe_rounds"], "!" if desc["streams_share_a_queue"] else ""))
print(" ".join(out))
