"""Where does the span kernel lose rate with stream size (const 5001 Hz 80 / 79.6 / 76 % at 1 / 2.3 / 4 GiB in)?

One const-mode stream of N = 2^LG samples (default 30) on the span kernel, timed three ways in one process, round-robin:
  whole     one plan, one launch over all N samples
  sub       the same samples as N / 2^28 plans of 2^28 samples run back to back (counter seeded from the closed form) —
            what sub-launches of <= 1 GiB inside run_plan would do
  part k    the k-th of those plans alone (same buffers, offset k GiB): is it the place in memory or the size of the launch?
MEM=torch (default) takes the buffers from torch's allocator, MEM=hip straight from hipMalloc.
    python tools/size_probe.py [shift_hz] [LG] [pair]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import doppler_amd  # noqa: E402
from doppler_amd import shard  # noqa: E402

RATE = 1024000
shift = float(sys.argv[1]) if len(sys.argv) > 1 else 5001.0
LG = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pair = sys.argv[3] if len(sys.argv) > 3 else "i16:i16"
it, ot = pair.split(":")
BPS = {"i16": 4, "f32": 8}
N = 1 << LG
SUB = 1 << int(os.environ.get("SUBLG", "28"))
ctx = doppler_amd.Context(0)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
if os.environ.get("MEM", "torch") == "hip":
    p_in, p_out = ctx.malloc(N * BPS[it]), ctx.malloc(N * BPS[ot])
    torch.cuda.synchronize()
else:
    x = torch.randint(-23170, 23171, (N * BPS[it] // 2,), dtype=torch.int16, device=dev)
    out = torch.empty(N * BPS[ot], dtype=torch.uint8, device=dev)
    p_in, p_out = x.data_ptr(), out.data_ptr()
whole = ctx.plan_const(shift, RATE, N)
parts = []
for k in range(N // SUB):
    parts.append(ctx.plan_const(shift, RATE, SUB, samplenum=shard.chunk_seed(shift, RATE, k * SUB)))


def run_part(k):
    parts[k].run(p_in + k * SUB * BPS[it], it, p_out + k * SUB * BPS[ot], ot, st.cuda_stream)


cases = {"whole": (lambda: whole.run(p_in, it, p_out, ot, st.cuda_stream), N),
         "sub": (lambda: [run_part(k) for k in range(len(parts))], N)}
for k in range(len(parts)):
    cases["part %d" % k] = ((lambda k=k: run_part(k)), SUB)
ms = {c: [] for c in cases}
for c, (fn, n) in cases.items():
    for _ in range(5):
        fn()
st.synchronize()
for rnd in range(7):
    for c, (fn, n) in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10 if n == N else 20
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        st.synchronize()
        ms[c].append(e0.elapsed_time(e1) / reps)
print("shift %g Hz, 2^%d samples, %s, buffers from %s, in at 0x%x out at 0x%x" % (shift, LG, pair, os.environ.get("MEM", "torch"), p_in, p_out))
for c, (fn, n) in cases.items():
    m = statistics.median(ms[c])
    print("%-8s %9.1f us  %5.1f %% of 8 TB/s" % (c, m * 1e3, n * (BPS[it] + BPS[ot]) / m / 1e6 / 80))
