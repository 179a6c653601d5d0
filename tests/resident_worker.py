"""One process of tests/test_gpu_resident.py::test_four_processes_share_the_gpu_with_resident_kernels: a stream of 8 KiB
blocks through dpx_shift_block_async / dpx_wait on a context of its own, checked against the oracle.  Prints `ok <seed>`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import doppler_amd  # noqa: E402
from doppler_amd import dsp  # noqa: E402
from helpers import assert_same_bytes, make_iq  # noqa: E402
from oracle import oracle as orc  # noqa: E402

seed, n_blocks = int(sys.argv[1]), int(sys.argv[2])
pairs = [("i16", "i16"), ("f32", "i16"), ("i16", "f32"), ("f32", "f32")]
it, ot = pairs[seed % 4]
shift, rate = [5001, -5234, 9876, 1234][seed % 4], 1024000       # `doppler const --shift` is an i32 (main.rs:110)
spb = 8192 // (4 if it == "i16" else 8)
x = make_iq(it, spb * n_blocks, 9000 + seed, full_scale=(it == "i16"))
want, sn_want = orc.const_stream(x, it, ot, shift, rate)
ctx = doppler_amd.Context(0)
sn, out, tickets = 0, [], []
for b in range(n_blocks):
    tk, sn = dsp.shift_block_async(x[b * 8192:(b + 1) * 8192], it, ot, sn, float(shift), rate, ctx=ctx)
    tickets.append(tk)
    if len(tickets) == 4:
        out.append(dsp.wait(tickets.pop(0), ot, ctx=ctx))
while tickets:
    out.append(dsp.wait(tickets.pop(0), ot, ctx=ctx))
assert sn == sn_want
assert_same_bytes(np.concatenate(out), want, ot, "process %d" % seed)
info = ctx.resident_info()
assert info["blocks"] == n_blocks and info["launches"] == info["stops"] + info["idle_exits"] + info["running"], info
ctx.close()
print("ok %d %s" % (seed, info))
