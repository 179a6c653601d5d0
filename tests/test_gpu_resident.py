"""GPU: the block-per-call path behind dpx_shift_block / dpx_shift_block_async / dpx_wait — the reference's own call shape,
one 8 KiB block per call (src/main.rs:113-118) — served by a RESIDENT kernel (doppler_amd/csrc/dpx_resident.cpp).
Every comparison is exact equality of output bytes with the oracle.  What is asserted about the kernel's life are
invariants of dpx_resident_info that hold on a loaded box as on an idle one (launches == stops + idle_exits + running;
blocks == blocks handed over), never a launch count that depends on how fast the host loop happens to run."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from helpers import BPS, assert_same_bytes, load_golden, make_iq
from test_gpu_parity import run_bulk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def consistent(info):
    assert info["launches"] == info["stops"] + info["idle_exits"] + info["running"], info
    assert info["launches"] <= info["blocks"] + info["stops"] + 1, info          # a launch is only ever caused by a block (or follows a stop)
    return info


def test_resident_block_kernel_lifecycle(orc):
    """Blocks really go through doorbells and one launch serves many; the kernel leaves by itself when idle (2 ms) and the next
    block starts it again; a change of format pair, a bulk plan, a table-bound block and dpx_synchronize in between each make
    it leave and come back; resident mode off gives the same bytes through a launch per block; and the context can be
    destroyed while the kernel is resident.  Bytes and counters against the oracle."""
    import doppler_amd
    from doppler_amd import dsp
    c2 = doppler_amd.Context(0)
    try:
        rate = 1024000
        x = make_iq("i16", 2048 * 600, 4242, full_scale=True)
        want, sn_want = orc.const_stream(x, "i16", "i16", 5001, rate)

        def run(pause_every=0, depth=4):
            sn, out, tickets = 0, [], []
            for b in range(600):
                tk, sn = dsp.shift_block_async(x[b * 8192:(b + 1) * 8192], "i16", "i16", sn, 5001.0, rate, ctx=c2)
                tickets.append(tk)
                if len(tickets) == depth:
                    out.append(dsp.wait(tickets.pop(0), "i16", ctx=c2))
                if pause_every and b % pause_every == pause_every - 1:
                    time.sleep(0.02)                     # ten idle periods of the GPU's own clock: the kernel has left, tickets outstanding
            while tickets:
                out.append(dsp.wait(tickets.pop(0), "i16", ctx=c2))
            assert sn == sn_want
            return np.concatenate(out)

        i0 = consistent(c2.resident_info())
        assert i0["launches"] == 0 and i0["running"] == 0
        assert_same_bytes(run(), want, "i16", "resident kernel, 4 in flight")
        i1 = consistent(c2.resident_info())
        assert i1["blocks"] - i0["blocks"] == 600 and i1["launches"] >= 1 and i1["tickets_in_flight"] == 0
        assert_same_bytes(run(pause_every=100), want, "i16", "resident kernel, idle pauses")
        i2 = consistent(c2.resident_info())
        # it left during each of the six pauses (the GPU's wall clock, not the host's) and a block brought it back
        assert i2["blocks"] - i1["blocks"] == 600 and i2["idle_exits"] - i1["idle_exits"] >= 5, (i1, i2)
        # the synchronous operator takes the same road; other work of the context in between makes the kernel leave first
        sn = 0
        got = []
        xf = make_iq("f32", 1024 * 8, 4243)
        wf, _ = orc.const_stream(xf, "f32", "i16", 5001, rate)
        for b in range(40):
            o, _, sn = dsp.shift_block(x[b * 8192:(b + 1) * 8192], "i16", "i16", sn, 5001.0, rate, ctx=c2)
            got.append(o)
            if b == 10:                                  # a bulk plan on the same context
                gb, fin = run_bulk(c2, x[:8192 * 64], "i16", "i16", [(2048 * 64, 5001.0)], rate)
                assert_same_bytes(gb, want[:8192 * 64], "i16", "bulk plan between resident blocks")
                after = consistent(c2.resident_info())
                assert after["running"] == 0               # it left before the plan's launch
            if b == 20:                                  # another format pair: another kernel
                snf, gf = 0, []
                for k in range(8):
                    o2, _, snf = dsp.shift_block(xf[k * 8192:(k + 1) * 8192], "f32", "i16", snf, 5001.0, rate, ctx=c2)
                    gf.append(o2)
                assert_same_bytes(np.concatenate(gf), wf, "i16", "f32 -> i16 blocks in between")
            if b == 30:
                c2.synchronize()
                assert c2.resident_info()["running"] == 0
                w0, _, _, _ = orc.shift_block(x[:8192], "i16", "i16", 7, 0.0, rate)     # shift 0: a table-bound block
                o3, _, _ = dsp.shift_block(x[:8192], "i16", "i16", 7, 0.0, rate, ctx=c2)
                assert_same_bytes(o3, w0, "i16", "table-bound block between resident blocks")
        assert_same_bytes(np.concatenate(got), want[:8192 * 40], "i16", "synchronous blocks through the resident kernel")
        i3 = consistent(c2.resident_info())
        assert i3["blocks"] - i2["blocks"] == 48 and i3["launches"] - i2["launches"] >= 4, (i2, i3)
        # resident mode off: a launch per block, the same bytes
        c2.set_resident(False)
        assert_same_bytes(run(), want, "i16", "a launch per block")
        i4 = consistent(c2.resident_info())
        assert (i4["launches"], i4["blocks"], i4["running"]) == (i3["launches"], i3["blocks"], 0)
        c2.set_resident(True)
        tk, _ = dsp.shift_block_async(x[:8192], "i16", "i16", 0, 5001.0, rate, ctx=c2)     # leave a ticket and the kernel behind
        assert tk != 0 and c2.resident_info()["tickets_in_flight"] == 1
    finally:
        c2.close()                                       # destroys the context while the kernel is resident


def test_tickets_of_two_kernel_instances_in_flight_together(orc):
    """A ticket is only ever served by the kernel instance it was rung for.  Asynchronous blocks of one format pair are left
    in flight while blocks of ANOTHER pair (and of the other libm build) are issued on the same context: the old instance
    serves what it was handed before the new one starts.  (a rung-but-unserved ticket could be picked up by a
    kernel of the wrong format after the idle clock.)  The pauses let the kernel idle out with tickets outstanding."""
    import doppler_amd
    from doppler_amd import dsp
    c = doppler_amd.Context(0)
    try:
        rate = 1024000
        xa = make_iq("i16", 2048 * 64, 11, full_scale=True)
        xb = make_iq("f32", 1024 * 64, 12)
        wa, _ = orc.const_stream(xa, "i16", "i16", 5001, rate)
        wb, _ = orc.const_stream(xb, "f32", "f32", -5234, rate)
        sna = snb = 0
        ga, gb = [], []
        for rnd in range(32):
            ta = []
            for k in range(2):
                b = rnd * 2 + k
                tk, sna = dsp.shift_block_async(xa[b * 8192:(b + 1) * 8192], "i16", "i16", sna, 5001.0, rate, ctx=c)
                ta.append(tk)
            if rnd % 8 == 3:
                time.sleep(0.01)                         # the i16 -> i16 kernel idles out; its two tickets are served, not waited for
            tb = []
            for k in range(2):                           # two more slots: another pair while the first two tickets are outstanding
                b = rnd * 2 + k
                tk, snb = dsp.shift_block_async(xb[b * 8192:(b + 1) * 8192], "f32", "f32", snb, -5234.0, rate, ctx=c)
                tb.append(tk)
            for tk in ta:
                ga.append(dsp.wait(tk, "i16", ctx=c))
            for tk in tb:
                gb.append(dsp.wait(tk, "f32", ctx=c))
        assert_same_bytes(np.concatenate(ga), wa, "i16", "i16 -> i16 tickets around f32 -> f32 ones")
        assert_same_bytes(np.concatenate(gb), wb, "f32", "f32 -> f32 tickets around i16 -> i16 ones")
        info = consistent(c.resident_info())
        assert info["blocks"] == 128 and info["launches"] >= 64, info       # every change of pair is a stop and a launch
    finally:
        c.close()


def test_two_contexts_on_one_device_interleave_async_blocks(orc):
    """Two contexts on device 0, their asynchronous blocks interleaved by one host thread: different shifts, different format
    pairs, four tickets in flight each.  There is ONE resident kernel per device and process (csrc/dpx_internal.h,
    DeviceState): HIP may map the two contexts' streams to one hardware queue, where two resident kernels take turns at the
    pace of the 2 ms idle clock (measured before the rule: 400 + 400 blocks in 1.6 s) — so the contexts hand the kernel
    over instead, block by block here, which is the worst case.  Both streams are the oracle's; the hand-over is bounded."""
    import doppler_amd
    from doppler_amd import dsp
    ca, cb = doppler_amd.Context(0), doppler_amd.Context(0)
    try:
        rate, nb = 1024000, 400
        xa = make_iq("i16", 2048 * nb, 21, full_scale=True)
        xb = make_iq("f32", 1024 * nb, 22)
        wa, sa = orc.const_stream(xa, "i16", "i16", 5001, rate)
        wb, sb = orc.const_stream(xb, "f32", "i16", -5234, rate)
        sna = snb = 0
        ga, gb, ta, tb = [], [], [], []
        t0 = time.perf_counter()
        for b in range(nb):
            tk, sna = dsp.shift_block_async(xa[b * 8192:(b + 1) * 8192], "i16", "i16", sna, 5001.0, rate, ctx=ca)
            ta.append(tk)
            tk, snb = dsp.shift_block_async(xb[b * 8192:(b + 1) * 8192], "f32", "i16", snb, -5234.0, rate, ctx=cb)
            tb.append(tk)
            if len(ta) == 4:
                ga.append(dsp.wait(ta.pop(0), "i16", ctx=ca))
                gb.append(dsp.wait(tb.pop(0), "i16", ctx=cb))
            ia, ib = ca.resident_info(), cb.resident_info()
            assert ia["running"] + ib["running"] <= 1, (ia, ib)          # never two resident kernels on the device
        while ta:
            ga.append(dsp.wait(ta.pop(0), "i16", ctx=ca))
            gb.append(dsp.wait(tb.pop(0), "i16", ctx=cb))
        dt = time.perf_counter() - t0
        assert (sna, snb) == (sa, sb)
        assert_same_bytes(np.concatenate(ga), wa, "i16", "context A")
        assert_same_bytes(np.concatenate(gb), wb, "i16", "context B")
        ia, ib = consistent(ca.resident_info()), consistent(cb.resident_info())
        assert ia["blocks"] == nb and ib["blocks"] == nb
        print("two contexts: %d + %d blocks in %.3f s (%.1f us per block); launches %d / %d, idle exits %d / %d" % (
            nb, nb, dt, dt / (2 * nb) * 1e6, ia["launches"], ib["launches"], ia["idle_exits"], ib["idle_exits"]))
        # handed over, not idled out: had every block waited for the other kernel's idle clock, 800 x 2 ms = 1.6 s
        assert ia["idle_exits"] + ib["idle_exits"] < nb // 4 or dt < 0.8, (dt, ia, ib)
        # a stretch of blocks on ONE context afterwards keeps its kernel: no hand-over without a second user
        l0 = ca.resident_info()["launches"]
        sn = 0
        for b in range(50):
            o, _, sn = dsp.shift_block(xa[b * 8192:(b + 1) * 8192], "i16", "i16", sn, 5001.0, rate, ctx=ca)
            assert_same_bytes(o, wa[b * 8192:(b + 1) * 8192], "i16", "context A alone, block %d" % b)
        assert ca.resident_info()["launches"] - l0 <= 50 // 2             # (a loaded box may idle it out now and then)
    finally:
        ca.close()
        cb.close()


def test_four_processes_share_the_gpu_with_resident_kernels():
    """Four processes (the shape of `DPX_BENCH_SHARE_GPU` development runs, or four `doppler` commands on one GPU), each with
    its own context and its own resident kernel on device 0, at once: every stream is the oracle's (the workers check)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "resident_worker.py"), str(k), "300"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for k in range(4)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for k, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("ok %d" % k) in o, "process %d:\n%s" % (k, o[-2000:])


def test_async_blocks_whose_plan_wants_a_table(ctx, orc):
    """dpx_shift_block_async with shifts whose period is below 4 — shift 0 (the reference resets the counter on EVERY sample:
    period 1), samplerate / 2 (period 2), samplerate / 3 — and an ordinary shift between them: such a block's plan asks for
    a corrector table, which the asynchronous slot has no room for; it takes the synchronous path inside the call (round 3
    returned DPX_ERR_PLAN: a drop-in for the loop of main.rs:113-118 must take every input the loop takes).  Both block
    formats, tickets in flight across the fallback, counters carried, against the oracle block by block."""
    from doppler_amd import dsp
    rate = 48000
    shifts = [0.0, 5000.0, 24000.0, 0.0, 16000.0, -24000.0, 123.0, 0.0]
    for intype, outtype, per in (("i16", "i16", 2048), ("f32", "i16", 1024), ("i16", "f32", 2048)):
        x = make_iq(intype, per * len(shifts) - 100, 77)          # the last block is short
        bs = 8192
        sn, sn_w, tickets, got, want = 3, 3, [], [], []
        for b, hz in enumerate(shifts):
            blk = x[b * bs:(b + 1) * bs]
            tk, sn = dsp.shift_block_async(blk, intype, outtype, sn, hz, rate, ctx=ctx)
            w, _, _, sn_w = orc.shift_block(blk, intype, outtype, sn_w, hz, rate)
            assert sn == sn_w, (b, hz)
            want.append(w)
            tickets.append(tk)
            if len(tickets) == 3:
                got.append(dsp.wait(tickets.pop(0), outtype, ctx=ctx))
        while tickets:
            got.append(dsp.wait(tickets.pop(0), outtype, ctx=ctx))
        assert_same_bytes(np.concatenate(got), np.concatenate(want), outtype, "async blocks with tiny periods %s->%s" % (intype, outtype))


def test_async_blocks_equal_the_synchronous_path_on_the_golden_track_replay(ctx, orc):
    """dpx_shift_block_async / dpx_wait (the loop of main.rs:113-118 with block k + 1 read while block k is on the GPU): the
    golden track replay block by block with two, then four blocks in flight — bytes and counters of the synchronous path
    and of the golden file; a fifth outstanding ticket, a stale ticket and a short output buffer are refused."""
    from doppler_amd import dsp
    from doppler_amd.engine import DspError
    t = load_golden("track_stream_case.npz")
    rate = int(t["meta"][0])
    x, want, log = t["x"], t["y"], t["shift_log"].astype(np.float32)
    nb = (x.size + 8191) // 8192
    assert nb == log.size
    for depth in (2, 4):
        sn, out, tickets = 0, [], []
        for b in range(nb):
            blk = x[b * 8192:(b + 1) * 8192]
            tk, sn = dsp.shift_block_async(blk, "i16", "i16", sn, float(log[b]), rate, ctx=ctx)
            tickets.append(tk)
            if len(tickets) == depth:
                out.append(dsp.wait(tickets.pop(0), "i16", ctx=ctx))
        while tickets:
            out.append(dsp.wait(tickets.pop(0), "i16", ctx=ctx))
        assert_same_bytes(np.concatenate(out), want, "i16", "async blocks, %d in flight" % depth)
        sn_sync = 0
        for b in range(nb):
            _, _, sn_sync = dsp.shift_block(x[b * 8192:(b + 1) * 8192], "i16", "i16", sn_sync, float(log[b]), rate, ctx=ctx)
        assert sn == sn_sync
    # f32 -> f32 and an empty block
    xf = make_iq("f32", 1024, 5)
    tk, sn = dsp.shift_block_async(xf, "f32", "f32", 7, -15000.0, 256000, ctx=ctx)
    tk0, sn0 = dsp.shift_block_async(xf[:0], "f32", "f32", sn, -15000.0, 256000, ctx=ctx)
    w, _, _, sn_w = orc.shift_block(xf, "f32", "f32", 7, -15000.0, 256000)
    assert_same_bytes(dsp.wait(tk, "f32", ctx=ctx), w, "f32", "async f32 block")
    assert dsp.wait(tk0, "f32", ctx=ctx).size == 0 and sn0 == sn == sn_w
    # misuse
    ts = [dsp.shift_block_async(xf, "f32", "f32", 0, 1.0, 48000, ctx=ctx)[0] for _ in range(4)]
    with pytest.raises(DspError):
        dsp.shift_block_async(xf, "f32", "f32", 0, 1.0, 48000, ctx=ctx)
    with pytest.raises(DspError):
        dsp.wait(ts[0] + 100, "f32", ctx=ctx)
    for tk in ts:
        dsp.wait(tk, "f32", ctx=ctx)
    with pytest.raises(DspError):
        dsp.wait(ts[0], "f32", ctx=ctx)
