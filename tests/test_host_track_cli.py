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


def test_sdp4_published_test_case():
    """NORAD SDP4 (deep-space) test case of Spacetrack Report #3 (satellite 11801, period 630 min, e = 0.73): the published
    state vectors at 0 / 360 / 720 / 1080 / 1440 minutes (single precision in the report: hence the tolerance).  The
    reference hands ANY element set to libgpredict (src/main.rs:141-149,162), which takes this model for periods of 225
    minutes and more; rounds 1-3 refused such sets.  Orbit parity with libgpredict itself stays unpinned."""
    d1 = b"1 11801U          80230.29629788  .01431103  00000-0  14311-1       8"
    d2 = b"2 11801  46.7916 230.4354 7318036  47.4722  10.4117  2.28537848     6"
    want = {
        0: (7473.37066650, 428.95261765, 5828.74786377, 5.10715413, 6.44468284, -0.18613096),
        360: (-3305.22537232, 32410.86328125, -24697.17675781, -1.30113538, -1.15131518, -0.28333528),
        720: (14271.28759766, 24110.46411133, -4725.76837158, -0.32050445, 2.67984074, -2.08405289),
        1080: (-9990.05883789, 22717.35522461, -23616.89062501, -1.01667246, -2.29026759, 0.72892364),
        1440: (9787.86975097, 33753.34667969, -15030.81176758, -1.09425066, 0.92358845, -1.52230928),
    }
    out = (C.c_double * 6)()
    for ts, w in want.items():
        assert doppler_amd.lib.dpx_orbit_propagate(d1, d2, float(ts), out) == 0
        for k in range(3):
            assert abs(out[k] - w[k]) < 0.02, (ts, k, out[k], w[k])            # km
            assert abs(out[3 + k] - w[3 + k]) < 2e-5, (ts, k, out[3 + k], w[3 + k])   # km/s


def test_sdp4_resonant_orbits_behave():
    """The report's one deep-space vector is not a resonant orbit; the two resonance branches (24-hour synchronous: three
    tesseral terms; 12-hour with e >= 0.5: ten terms, both integrated in 720-minute steps from the epoch) are checked for what
    must hold whatever the coefficients: a geostationary satellite stays at the geostationary radius and over its longitude for
    days, a Molniya orbit keeps its perigee and apogee radii and its 718-minute period, the state is continuous across the
    integrator's step boundaries and forwards / backwards in time, and velocity is the derivative of position."""
    out, o2 = (C.c_double * 6)(), (C.c_double * 6)()
    lib = doppler_amd.lib
    geo = (b"1 99991U 20001A   20001.00000000  .00000000  00000-0  00000-0 0  9991",
           b"2 99991   0.0500  80.0000 0002000  40.0000 200.0000  1.00273000    17")
    mol = (b"1 99992U 20002A   20001.00000000  .00000000  00000-0  00000-0 0  9992",
           b"2 99992  63.4000 120.0000 7200000 270.0000  10.0000  2.00600000    18")
    r = lambda o: float(np.sqrt(o[0] ** 2 + o[1] ** 2 + o[2] ** 2))
    # geostationary: radius and earth-fixed longitude over four days (mean motion 1.00273 rev/day = one per sidereal day)
    lons = []
    for ts in np.arange(-1440.0, 4 * 1440.0 + 1, 180.0):
        assert lib.dpx_orbit_propagate(geo[0], geo[1], float(ts), out) == 0
        assert abs(r(out) - 42164.0) < 60.0, (ts, r(out))
        lons.append(np.degrees(np.arctan2(out[1], out[0])) - 360.0 * 1.00273790934 * ts / 1440.0)
    lons = np.unwrap(np.radians(lons))
    assert np.ptp(lons) < np.radians(1.5), np.degrees(np.ptp(lons))
    # Molniya: perigee / apogee radii from a = (mu / n^2)^(1/3), period 1440 / 2.006 minutes
    a = (398600.8 / (2.006 * 2 * np.pi / 86400.0) ** 2) ** (1.0 / 3.0)
    rs = []
    ts_all = np.arange(0.0, 3 * 1440.0, 2.0)
    for ts in ts_all:
        assert lib.dpx_orbit_propagate(mol[0], mol[1], float(ts), out) == 0
        rs.append(r(out))
    rs = np.array(rs)
    assert abs(rs.min() - a * (1 - 0.72)) < 80.0 and abs(rs.max() - a * (1 + 0.72)) < 300.0, (rs.min(), rs.max(), a)
    perigees = ts_all[1:-1][(rs[1:-1] < rs[:-2]) & (rs[1:-1] < rs[2:])]
    assert len(perigees) >= 5 and np.allclose(np.diff(perigees), 1440.0 / 2.006, atol=3.0), perigees
    for tle in (geo, mol):
        # continuity across the integrator's steps (720, 1440 min ...), both directions, and velocity = d position / dt
        for ts in (719.999, 720.0, 1439.9995, 2160.0, -719.9995, -720.0, 5.0, 3000.0):
            assert lib.dpx_orbit_propagate(tle[0], tle[1], float(ts), out) == 0
            assert lib.dpx_orbit_propagate(tle[0], tle[1], float(ts) + 0.002, o2) == 0        # 0.12 s later
            for k in range(3):
                assert abs((o2[k] - out[k]) / 0.12 - 0.5 * (out[3 + k] + o2[3 + k])) < 2e-3, (tle[1][2:7], ts, k)
    # what `doppler track` consumes: a range rate for a deep-space element set, equal to the finite difference of the range
    la, lb = (C.c_double * 4)(), (C.c_double * 4)()
    import calendar
    t0 = calendar.timegm((2020, 1, 1, 6, 0, 0))
    assert lib.dpx_orbit_observe(mol[0], mol[1], 58.26541, 26.46667, 76.0, float(t0), la) == 0
    assert lib.dpx_orbit_observe(mol[0], mol[1], 58.26541, 26.46667, 76.0, float(t0) + 1.0, lb) == 0
    assert abs((lb[2] - la[2]) - 0.5 * (la[3] + lb[3])) < 1e-4, (la[:], lb[:])


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
