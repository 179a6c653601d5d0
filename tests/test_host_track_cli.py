"""CPU: host side of track mode (schedule, orbit) and the CLI's argument surface.  No GPU needed:
argument errors are reported before any device is touched."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import doppler_amd
from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "doppler_amd", "bin", "doppler")


def track_schedule(rr, rate, freq, offset, in_fmt, nbytes):
    lib = doppler_amd.lib
    rr = np.ascontiguousarray(rr, dtype=np.float64)
    cap = nbytes // 8192 + 2
    out = np.empty(cap, dtype=np.float32)
    nb = C.c_size_t()
    rc = lib.dpx_track_schedule(rr.ctypes.data, rr.size, rate, freq, 0 if offset is None else offset,
                                0 if offset is None else 1, in_fmt, nbytes, out.ctypes.data, cap, C.byref(nb))
    assert rc == 0, lib.dpx_last_error()
    return out[: nb.value]


def test_track_schedule_matches_reference_loop(orc):
    """dpx_track_schedule vs the oracle's restatement of main.rs:156-184 (golden log and fresh cases):
    one-block lag, whole seconds truncated through f32, f32 offset add."""
    t = load_golden("track_stream_case.npz")
    rate, freq, off, _ = t["meta"]
    got = track_schedule(t["rr"], int(rate), int(freq), int(off), 0, t["x"].size)
    assert np.array_equal(got, t["shift_log"])
    rng = np.random.default_rng(5)
    for rate, fmt, name in [(256000, 0, "i16"), (48000, 1, "f32"), (2400000, 0, "i16")]:
        rr = rng.uniform(-7.5, 7.5, size=9)
        nbytes = 8192 * 300 + (8 if fmt else 4) * 17
        x = np.zeros(nbytes, dtype=np.uint8)
        for offset in (None, -2500, 5000):
            _, _, log = orc.track_stream(x, name, name, rate, 437505000, rr, offset_hz=offset)
            got = track_schedule(rr, rate, 437505000, offset, fmt, nbytes)
            assert np.array_equal(got, log), (rate, name, offset)


def test_sgp4_published_test_case():
    """NORAD SGP4 test case of Spacetrack Report #3 (satellite 88888): the published state vectors,
    which were computed in single precision — hence the tolerance.  Orbit parity with libgpredict
    itself is unpinned (the library is not part of the reference tree)."""
    l1 = b"1 88888U          80275.98708465  .00073094  13844-3  66816-4 0    87"
    l2 = b"2 88888  72.8435 115.9689 0086731  52.6988 110.5714 16.05824518  1058"
    want = {
        0: (2328.97048951, -5995.22076416, 1719.97067261, 2.91207230, -0.98341546, -7.09081703),
        360: (2456.10705566, -6071.93853760, 1222.89727783, 2.67938992, -0.44829041, -7.22879231),
        720: (2567.56195068, -6112.50384522, 713.96397400, 2.44024599, 0.09810869, -7.31995916),
        1080: (2663.09078980, -6115.48229980, 196.39640427, 2.19611958, 0.65241995, -7.36282432),
        1440: (2742.55133057, -6079.67144775, -326.38095856, 1.94850229, 1.21106251, -7.35619372),
    }
    out = (C.c_double * 6)()
    for ts, w in want.items():
        assert doppler_amd.lib.dpx_orbit_propagate(l1, l2, float(ts), out) == 0
        for k in range(3):
            assert abs(out[k] - w[k]) < 0.02, (ts, k, out[k], w[k])            # km
            assert abs(out[3 + k] - w[3 + k]) < 2e-5, (ts, k, out[3 + k], w[3 + k])   # km/s
    # deep-space sets are rejected, not silently mis-propagated
    d2 = b"2 11801  46.7916 230.4354 7318036  47.4722  10.4117  2.28537848    13"
    d1 = b"1 11801U          80230.29629788  .01431103  00000-0  14311-1       8"
    assert doppler_amd.lib.dpx_orbit_propagate(d1, d2, 0.0, out) != 0


def test_orbit_observe_is_physically_consistent():
    """Range rate equals the finite difference of range; elevation/azimuth in range; LEO speeds."""
    l1 = b"1 88888U          80275.98708465  .00073094  13844-3  66816-4 0    87"
    l2 = b"2 88888  72.8435 115.9689 0086731  52.6988 110.5714 16.05824518  1058"
    import calendar
    epoch = calendar.timegm((1980, 1, 1, 0, 0, 0)) + (275.98708465 - 1.0) * 86400.0
    out = (C.c_double * 4)()
    lib = doppler_amd.lib
    for t in np.linspace(0, 5400, 19):
        a = []
        for dt in (-0.5, 0.0, 0.5):
            assert lib.dpx_orbit_observe(l1, l2, 58.26541, 26.46667, 76.0, epoch + t + dt, out) == 0
            a.append(list(out))
        az, el, rng_km, rr = a[1]
        assert 0 <= az < 360 and -90 <= el <= 90 and 100 < rng_km < 14000 and abs(rr) < 8.5
        fd = (a[2][2] - a[0][2]) / 1.0
        assert abs(fd - rr) < 2e-3, (t, fd, rr)


def run_cli(args, stdin=b""):
    return subprocess.run([EXE] + args, input=stdin, capture_output=True, timeout=60)


def test_cli_argument_surface():
    """usage.rs:117-337: subcommands, required flags, possible values, leading-hyphen values, exit status 1."""
    assert os.path.exists(EXE), "build the CLI with `make cli`"
    r = run_cli([])
    assert r.returncode == 1 and b"no arguments provided, try with doppler -h" in r.stderr
    r = run_cli(["const", "-s", "1024000", "-i", "i16"])
    assert r.returncode == 1 and b"--shift <SHIFT>" in r.stderr
    r = run_cli(["const", "-s", "1024000", "-i", "i8", "--shift", "5"])
    assert r.returncode == 1 and b"isn't a valid value" in r.stderr and b"i16, f32" in r.stderr
    r = run_cli(["const", "-s", "-5", "-i", "i16", "--shift", "5"])
    assert r.returncode == 1 and b"isn't a valid value" in r.stderr
    r = run_cli(["const", "-s", "1024000", "-i", "i16", "--shift", "5.5"])
    assert r.returncode == 1
    r = run_cli(["bogus"])
    assert r.returncode == 1
    r = run_cli(["const", "--help"])
    assert r.returncode == 0 and b"--samplerate" in r.stdout and b"--shift" in r.stdout
    r = run_cli(["track", "-s", "256000", "-i", "i16", "--tlefile", "x", "--tlename", "y", "--frequency", "437505000"])
    assert r.returncode == 1 and b"--location <LOCATION>" in r.stderr
    base = ["track", "-s", "256000", "-i", "i16", "--tlefile", "/nonexistent", "--tlename", "ESTCUBE 1",
            "--frequency", "437505000"]
    r = run_cli(base + ["--location", "lat=58.2,lon=26.4"])
    assert r.returncode == 1 and b"--location should be defined as: lat=58.64560,lon=23.15163,alt=8" in r.stderr
    r = run_cli(base + ["--location", "lat=abc,lon=26.4,alt=7"])
    assert r.returncode == 1 and b"isn't a valid value for --location" in r.stderr
    r = run_cli(base + ["--location", "lat=58.2,lon=26.4,alt=76", "--time", "2015-13-40"])
    assert r.returncode == 1 and b"--time should be defined in Y-m-dTH:M:S format" in r.stderr
    # negative values after a flag parse as values (AllowLeadingHyphen, usage.rs:127,161); the run then stops
    # at the missing TLE file (status 1, like the reference's exit(1) at main.rs:145) or, on a box without
    # a GPU, never gets that far for `const`
    r = run_cli(base + ["--location", "lat=58.2,lon=26.4,alt=76", "--offset", "-2500", "--time", "2015-01-22T09:07:16"])
    assert r.returncode == 1 and b"cannot open TLE file" in r.stderr
