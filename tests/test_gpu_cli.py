"""GPU: the `doppler` command end to end (stdin -> GPU -> stdout) against the oracle's restatement of the
reference driver loops (main.rs:102-119 const, main.rs:156-184 track replay)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from helpers import BPS, assert_same_bytes, make_iq

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "doppler_amd", "bin", "doppler")


def run_cli(args, data, env=None):
    e = dict(os.environ)
    if env:
        e.update(env)
    return subprocess.run([EXE] + args, input=bytes(data), capture_output=True, timeout=300, env=e)


@pytest.mark.parametrize("intype,outtype,shift,rate", [("i16", "i16", 5000, 1024000), ("f32", "i16", -15000, 256000),
                                                       ("i16", "f32", 815000, 2400000), ("f32", "f32", 12345, 1024000)])
def test_const_mode_end_to_end(orc, intype, outtype, shift, rate):
    n = 2048 * 700 + 123          # ragged but whole-sample tail
    x = make_iq(intype, n, 3, full_scale=True)
    want, _ = orc.const_stream(x, intype, outtype, shift, rate, threads=4)
    want1, _ = orc.const_stream(x[: 8192 * 3], intype, outtype, shift, rate)
    args = ["const", "-s", str(rate), "-i", intype, "--shift", str(shift)] + (["-o", outtype] if outtype != intype else [])
    for slab in ("33554432", "65536", "8192"):      # one slab, many slabs, one reference block per launch
        r = run_cli(args, x, {"DOPPLER_SLAB_BYTES": slab})
        assert r.returncode == 0, r.stderr[-500:]
        got = np.frombuffer(r.stdout, dtype=np.uint8)
        assert_same_bytes(got, want, outtype, "slab %s" % slab)
    r = run_cli(args, x[: 8192 * 3])                # exact multiple of the block size: ends on the empty read
    assert r.returncode == 0
    assert_same_bytes(np.frombuffer(r.stdout, dtype=np.uint8), want1, outtype, "3 blocks")
    assert b"constant shift mode" in r.stderr and b"frequency shift" in r.stderr


def test_trailing_partial_sample_aborts_like_the_reference(orc):
    """8192*k + 10 bytes of i16: the reference asserts on the last block (dsp.rs:87) after having written the
    complete blocks; nothing of the ragged block is produced and the status is 101 (Rust panic)."""
    x = make_iq("i16", 2048 * 5 + 3, 9)[: 8192 * 5 + 10]
    want, _ = orc.const_stream(x[: 8192 * 5], "i16", "i16", 777, 48000)
    r = run_cli(["const", "-s", "48000", "-i", "i16", "--shift", "777"], x)
    assert r.returncode == 101 and b"assertion failed" in r.stderr
    assert_same_bytes(np.frombuffer(r.stdout, dtype=np.uint8), want, "i16", "complete blocks only")
    r = run_cli(["const", "-s", "48000", "-i", "i16", "--shift", "777"], b"")
    assert r.returncode == 0 and r.stdout == b""


def test_live_pipe_is_processed_in_small_steps(orc):
    """A slow producer (like rtl_fm): output must appear block by block, not after a slab has filled."""
    import time
    n = 2048 * 6
    x = make_iq("i16", n, 4)
    want, _ = orc.const_stream(x, "i16", "i16", 5000, 1024000)
    p = subprocess.Popen([EXE, "const", "-s", "1024000", "-i", "i16", "--shift", "5000"], stdin=subprocess.PIPE,
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    os.set_blocking(p.stdout.fileno(), False)
    got = b""
    seen_early = False
    for b in range(6):
        p.stdin.write(bytes(x[b * 8192:(b + 1) * 8192]))
        p.stdin.flush()
        t0 = time.time()
        while time.time() - t0 < 5.0 and len(got) < (b + 1) * 8192:
            chunk = p.stdout.read()
            if chunk:
                got += chunk
            else:
                time.sleep(0.01)
        if b < 5 and len(got) >= (b + 1) * 8192:
            seen_early = True
    p.stdin.close()
    p.wait(timeout=30)
    os.set_blocking(p.stdout.fileno(), True)
    got += p.stdout.read()
    assert seen_early, "no output before the input ended"
    assert_same_bytes(np.frombuffer(got, dtype=np.uint8), want, "i16", "live pipe")


def test_track_replay_with_range_rate_table(orc):
    """Track replay driven by a range-rate table (extension flag): the CLI's schedule + kernel vs the oracle."""
    rate, freq, off = 256000, 437505000, -2500
    rr = 6.8 * np.tanh((np.arange(16) - 6.0) / 2.0)
    n = rate * 9 + 2048 * 2 + 55
    x = make_iq("i16", n, 5)
    want, _, _ = orc.track_stream(x, "i16", "i16", rate, freq, rr, offset_hz=off)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write("\n".join("%.17g" % v for v in rr))
        path = f.name
    try:
        for slab in ("33554432", "262144"):
            r = run_cli(["track", "-s", str(rate), "-i", "i16", "--range-rate-file", path, "--frequency", str(freq),
                         "--offset", str(off), "--time", "2015-01-22T09:07:16"], x, {"DOPPLER_SLAB_BYTES": slab})
            assert r.returncode == 0, r.stderr[-400:]
            assert_same_bytes(np.frombuffer(r.stdout, dtype=np.uint8), want, "i16", "track table slab %s" % slab)
        assert b"tracking mode" in r.stderr and b"doppler@437.505 MHz" in r.stderr
    finally:
        os.unlink(path)


def test_track_replay_with_tle(orc):
    """Full `doppler track --tlefile ... --time ...` (README recipe): SGP4 range rates at whole seconds (checked
    separately against the published SGP4 test case) fed through the oracle's schedule give the expected output."""
    import calendar
    import doppler_amd
    l1 = "1 88888U          80275.98708465  .00073094  13844-3  66816-4 0    87"
    l2 = "2 88888  72.8435 115.9689 0086731  52.6988 110.5714 16.05824518  1058"
    rate, freq = 48000, 437505000
    t0 = calendar.timegm((1980, 10, 1, 23, 50, 0))
    out = (C.c_double * 4)()
    rr = []
    for dt in range(0, 12):
        assert doppler_amd.lib.dpx_orbit_observe(l1.encode(), l2.encode(), 58.26541, 26.46667, 76.0, float(t0 + dt), out) == 0
        rr.append(out[3])
    n = rate * 8 + 777
    x = make_iq("f32", n, 6)
    want, _, _ = orc.track_stream(x, "f32", "i16", rate, freq, np.array(rr), offset_hz=None)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write("OTHER SAT\n1 00000U 00000A   80275.98708465  .00000000  00000-0  00000-0 0    00\n"
                "2 00000  10.0000 000.0000 0000001  00.0000 000.0000 15.00000000    00\n")
        f.write("TEST SAT 88888  \n" + l1 + "\n" + l2 + "\n")
        path = f.name
    try:
        r = run_cli(["track", "-s", str(rate), "-i", "f32", "-o", "i16", "--tlefile", path, "--tlename", "TEST SAT 88888",
                     "--location", "lat=58.26541,lon=26.46667,alt=76", "--frequency", str(freq), "--time",
                     "1980-10-01T23:50:00"], x)
        assert r.returncode == 0, r.stderr[-400:]
        assert_same_bytes(np.frombuffer(r.stdout, dtype=np.uint8), want, "i16", "track tle")
        r = run_cli(["track", "-s", str(rate), "-i", "f32", "--tlefile", path, "--tlename", "NOT THERE",
                     "--location", "lat=58.26541,lon=26.46667,alt=76", "--frequency", str(freq)], x[:64])
        assert r.returncode == 1 and b"not found" in r.stderr
    finally:
        os.unlink(path)
