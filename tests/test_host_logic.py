"""CPU: the product's host-side logic (planner / closed-form counter / sharding) against the oracle's
sequential counter, and the C ABI surface.  No compute calls: there is no GPU here."""
import ctypes as C
import os

import numpy as np
import pytest

import doppler_amd
from doppler_amd import _lib, engine, shard
from helpers import oracle_counters

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RATIOS = [(5000.0, 1024000), (-15000.0, 256000), (815000.0, 2400000), (0.0, 1024000), (9876.543, 1024000),
          (-5234.17, 1024000), (3.0, 1024000), (1.0, 3), (7.0, 2), (123456.0, 48000)]


def test_library_exports_every_declared_symbol():
    """Three public headers: the boundary proper (doppler_hip.h: at most 30 entry points — what a binding of the path needs),
    the host-only callers' arithmetic (doppler_hip_host.h) and the measurement / self-check / helper surface
    (doppler_hip_debug.h).  Every declared function is exported and bound; nothing is exported beside them."""
    core = _lib.declared_symbols("doppler_hip.h")
    host = _lib.declared_symbols("doppler_hip_host.h")
    debug = _lib.declared_symbols("doppler_hip_debug.h")
    assert 25 <= len(core) <= 30, len(core)
    assert not (set(core) & set(host)) and not (set(core) & set(debug)) and not (set(host) & set(debug))
    for knob in ("dpx_set_options", "dpx_set_tuning", "dpx_plan_simulate", "dpx_plan_layout", "dpx_plan_describe", "dpx_debug_copy"):
        assert knob in debug and knob not in core
    declared = _lib.declared_symbols()
    assert set(declared) == set(core) | set(host) | set(debug)
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) == set(_lib._SIGNATURES), set(declared) ^ set(_lib._SIGNATURES)
    assert doppler_amd.lib.dpx_abi_version() == 5
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("dpx_")}
    assert exported == set(declared), exported ^ set(declared)
    assert os.path.getsize(_lib.LIB_PATH) < 4 * 1024 * 1024, "the library grew past 4 MB: which instantiations came back?"


def test_only_the_product_library_ships():
    """doppler_amd/lib travels to every GPU box with the tree: it holds libdoppler_hip.so and the objects it was linked from —
    no A/B leftovers (round 5 shipped two 3.9 MB alternates there), and __graft_entry__.build() leaves it that way.  Helper
    binaries of tools/ are built outside the tree (tools/build_tools.sh)."""
    libdir = os.path.join(ROOT, "doppler_amd", "lib")
    names = sorted(os.listdir(libdir))
    assert [n for n in names if n.endswith(".so")] == ["libdoppler_hip.so"], names
    assert all(n.endswith(".o") or n == "libdoppler_hip.so" for n in names), names
    assert not os.path.exists(os.path.join(ROOT, "tools", "bin")), "tools/bin is back: build helpers with tools/build_tools.sh (into /tmp)"
    import __graft_entry__
    __graft_entry__.build()
    assert sorted(os.listdir(libdir)) == names


def test_no_cpu_fallback_anywhere():
    """Without a usable GPU the product must fail loudly; and it must never import the oracle."""
    import sys
    n = C.c_int()
    rc = doppler_amd.lib.dpx_device_count(C.byref(n))
    if rc != 0 or n.value == 0:
        with pytest.raises(doppler_amd.DspError) as e:
            doppler_amd.Context(0)
        assert e.value.code == _lib.ERR_NO_DEVICE
        with pytest.raises(doppler_amd.DspError):
            doppler_amd.dsp.shift_block(np.zeros(8, np.uint8), "i16", "i16", 0, 1.0, 1000)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "doppler_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".cuh")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "liboracle", "oracle/", "orc_"):
                    assert needle not in text, "%s references the test oracle (%r)" % (os.path.join(dirpath, f), needle)
    assert "oracle" not in sys.modules or True   # the test session itself may have imported it; the package must not


@pytest.mark.parametrize("shift,rate", RATIOS)
def test_closed_form_counter_matches_sequential_rule(orc, shift, rate):
    n = 40000
    for sn0 in (0, 1, 5, 1023, 1024, 77777):
        want, sn_end = oracle_counters(orc, [(n, shift)], rate, sn0)
        stretches, fin = doppler_amd.plan_describe([(n, shift)], rate, sn0)
        assert fin == sn_end
        got = np.empty(n, np.uint32)
        pos = 0
        for s in stretches:
            assert s["first"] == pos
            j = np.arange(s["count"], dtype=np.uint64)
            if s["period"] == 0:
                got[pos:pos + s["count"]] = s["n_start"] + j
            else:
                got[pos:pos + s["count"]] = ((s["n_start"] - 1 + j) % s["period"]) + 1
            pos += s["count"]
        assert pos == n
        assert np.array_equal(got, want), (shift, rate, sn0, int(np.flatnonzero(got != want)[0]))
        assert engine.samplenum_after(shift, rate, sn0, n) == sn_end
        assert engine.samplenum_after(shift, rate, sn0, 0) == sn0


def test_find_reset_and_period(orc):
    for shift, rate, P in [(5000.0, 1024000, 1024), (-15000.0, 256000, 256), (815000.0, 2400000, 480),
                           (0.0, 1024000, 1), (9876.543, 1024000, 2592), (-5234.17, 1024000, 107405)]:
        assert engine.find_reset(shift, rate, 1, 1 << 22) == P
        assert engine.find_reset(shift, rate, 0, 10) == 0
        assert engine.find_reset(shift, rate, 1, P - 1) is None
    # 3 Hz at 1.024 Msps: the first reset from 1 is a full second away
    assert engine.find_reset(3.0, 1024000, 1, 2000000) == 1024000


def writes_ok(w, segs, rate, sn0, block, vecs, variant):
    """Every sample produced exactly once."""
    return bool((w == 1).all())


def test_launch_lists_cover_every_sample_once(orc):
    """dpx_plan_simulate mirrors the kernels' index arithmetic on the host: every sample written exactly
    once and with the counter value of the sequential rule — for the rows kernel, the tile kernel, the
    per-sample path, stretch boundaries and carried counters (track mode)."""
    cases = [
        ([(300000, 5000.0)], 1024000, 0),
        ([(131072 + 17, -15000.0)], 256000, 200),
        ([(200000, 815000.0), (150000, 9876.543), (1234, 3.0), (100000, 0.0), (70001, 5000.0)], 2400000, 0),
        ([(2048, 100.0), (2048, 101.5), (2048 * 40, 5000.0), (100, -3.25)], 1024000, 0),
        ([(1, 5000.0)], 1024000, 0), ([(255, 5000.0)], 1024000, 3), ([(70000, 1.0)], 3, 0),
    ]
    for segs, rate, sn0 in cases:
        want, _ = oracle_counters(orc, segs, rate, sn0)
        for variant in (3, 4, 1, 2, 5, 6):
            for block, vecs in ((256, 1), (128, 2)):
                c, w = doppler_amd.plan_simulate(segs, rate, sn0, block, vecs, variant)
                assert writes_ok(w, segs, rate, sn0, block, vecs, variant), (segs, variant, block, vecs, np.flatnonzero(w != 1)[:5])
                assert np.array_equal(c, want), (segs, variant, block, vecs, np.flatnonzero(c != want)[:5])


def test_rows_launches_that_evaluate_their_correctors(orc):
    """A rows launch of a long period carries its table AND (ratio, idx0), and launch_rows decides from the formats which
    to use; the host mirror checks that both name the same counters (0xfffffff9 otherwise) — for rows of one period, rows
    of several periods, a counter carried in, 4 and 8 rows per wavefront, the planner's own rule and the overrides."""
    cases = [([(1 << 21, 100.0)], 1024000, 0), ([(1 << 20, 9876.543)], 1024000, 0), ([(900000, 815000.0)], 2400000, 77),
             ([(1 << 21, 160.0)], 1024000, 3001), ([(600000, 250.0), (3000000, 50.0)], 1024000, 5)]
    for segs, rate, sn0 in cases:
        want, _ = oracle_counters(orc, segs, rate, sn0)
        for opts in (dict(), dict(rows_compute=1), dict(rows_compute=1, rows_r=4), dict(rows_compute=0xffffffff), dict(rows_compute=5000)):
            lay = doppler_amd.plan_layout(segs, rate, sn0, 128, 2, 6, options=opts)
            assert lay["rows_launches"] == len(segs), (segs, opts, lay)
            c, w = doppler_amd.plan_simulate(segs, rate, sn0, 128, 2, 6, options=opts)
            assert (w == 1).all() and np.array_equal(c, want), (segs, opts, np.flatnonzero(c != want)[:5])


def test_span_kernel_plans(orc):
    """Track-shaped plans (many constant-shift segments, counters carried across): more than eight periodic
    stretches sends the plan to the span kernel (one launch: matrices with 32-sample-aligned shifted rows, leftover
    blocks, tile launches for long uncovered gaps).  Index arithmetic checked against the sequential rule, for every
    format pair (a launch cuts its grid per pair: windows shared by two workgroups, one-matrix spans cut again)."""
    rng = np.random.default_rng(77)
    plans = [
        # twelve segments of arbitrary f32 shifts: odd periods, lead-ins where the carried counter exceeds the new period
        ([(120000 + 2048 * int(rng.integers(0, 9)), float(np.float32(rng.uniform(-9000, 9000)))) for _ in range(12)], 256000, 0),
        # short periods (row length is a multiple of the period), power-of-two and odd
        ([(30000, 1000.0 * (2 * k + 1)) for k in range(10)], 1024000, 0),
        ([(50000 + 17 * k, 333.0 + k) for k in range(10)], 48000, 5),
        # a long untabulated stretch in the middle becomes a tile launch; a short one a leftover range
        ([(30000, 500.0 + k) for k in range(9)] + [(90000, 0.001)] + [(3000, 0.002)] + [(30000, 700.0 + k) for k in range(3)], 64000, 0),
    ]
    for i, (segs, rate, sn0) in enumerate(plans):
        want, _ = oracle_counters(orc, segs, rate, sn0)
        lay = doppler_amd.plan_layout(segs, rate, sn0, 128, 2, 3)
        assert sum(lay[k] for k in ("rows_samples", "walk_samples", "tile_samples", "single_samples")) == lay["n_samples"]
        if i < 3:    # these really are span-kernel plans: matrices, leftover ranges, and (first plan) tile launches
            assert lay["walk_launches"] == 1 and lay["walk_matrices"] >= 8 and lay["leftover_ranges"] > 0, lay
            assert lay["rows_launches"] == 0, lay      # many tabulated stretches: never a rows plan, whatever their geometry
            # such a plan runs f32 -> i16 as ONE tile launch over the stream (launches_for): the simulation below walks that
            # for the pair under variant 3, and the span launch under variant 5 (the planner's choice overridden)
            assert lay["f32_i16_by_tiles"] == 1 and doppler_amd.plan_layout(segs, rate, sn0, 128, 2, 5)["f32_i16_by_tiles"] == 0, lay
        for variant in (3, 5):
            for pair in PAIRS:
                c, w = doppler_amd.plan_simulate(segs, rate, sn0, 128, 2, variant, pair=pair)
                assert (w == 1).all(), (segs[:3], variant, pair, np.flatnonzero(w != 1)[:5], w[np.flatnonzero(w != 1)[:5]])
                assert np.array_equal(c, want), (segs[:3], variant, pair, np.flatnonzero(c != want)[:5])
        # the measurement knobs (dpx_options) change the launch shapes, never the counters
        for opts in (dict(walk_waves=4), dict(walk_waves=8), dict(walk_waves=5, walk_tilemin=1000), dict(rows_r=4, rows_mult=3), dict(walk_waves=2),
                     dict(walk_span=2, walk_waves=2), dict(walk_span=5), dict(walk_span=64, walk_waves=8), dict(walk_span=4096, walk_waves=5),
                     dict(walk_flags=1), dict(walk_span=3, walk_flags=1)):
            for pair in (("i16", "i16"), ("f32", "f32")):
                c, w = doppler_amd.plan_simulate(segs, rate, sn0, 128, 2, 3, options=opts, pair=pair)
                assert (w == 1).all() and np.array_equal(c, want), (i, opts, pair)
        if i in (1, 2):    # no tile launch in these plans: no table at all (span workgroups evaluate their slices)
            assert doppler_amd.plan_layout(segs, rate, sn0, 128, 2, 3)["table_entries"] == 0


PAIRS = (("i16", "i16"), ("i16", "f32"), ("f32", "i16"), ("f32", "f32"))


def test_short_and_medium_matrices_share_workgroups_differently(orc):
    """Round 4: a span of up to 4 rows gives its workgroups 2 (up to 2 rows: 4) adjacent windows, WAVES / 2 (/ 4) wavefronts
    each.  Matrices of 2..13 rows of one period each
    (P = 8192 at 262 144 Hz: shifts that are odd multiples of 32 Hz), with and without a ragged last row, all format pairs,
    the launch's own 8 wavefronts for f32 -> i16 included; the layout's workgroup count shows the sharing."""
    rate, P = 262144, 8192
    for rows in range(2, 14):
        segs = []
        for k in range(9):         # nine matrices: a span plan, not a rows plan
            segs.append((rows * P + (k % 3) * 1000, 32.0 * (2 * k + 1)))
        want, _ = oracle_counters(orc, segs, rate, 0)
        lay = doppler_amd.plan_layout(segs, rate, 0, 128, 2, 3)
        assert lay["walk_launches"] == 1 and lay["walk_matrices"] >= 5 and lay["rows_launches"] == 0, (rows, lay)   # (a lead-in may leave a stretch under two rows)
        one = doppler_amd.plan_layout(segs, rate, 0, 128, 2, 3, options=dict(walk_span=16))["walk_workgroups"]   # one window per workgroup
        if rows <= 4:
            assert lay["walk_workgroups"] < one, (rows, lay["walk_workgroups"], one)      # windows shared: fewer workgroups
        for pair in PAIRS:
            c, w = doppler_amd.plan_simulate(segs, rate, 0, 128, 2, 3, pair=pair)
            assert (w == 1).all() and np.array_equal(c, want), (rows, pair, np.flatnonzero(w != 1)[:5], np.flatnonzero(c != want)[:5])


def test_one_matrix_launches_as_every_format_pair_cuts_them(orc):
    """Const mode on the span kernel: the launch cuts the spans of a one-matrix plan again for pairs with an f32 side
    (spans of 4 under 2, 4 or 5 wavefronts) — after planning.  The host mirror walks the grid each pair launches
    (span_launch_shape, the function the launch wrapper calls): row partition, leftover indexing, the 65535 grid-row limit."""
    for (shift, rate), sn0, n in (((5001.0, 1024000), 0, 3000000 + 77), ((777.0, 1024000), 12345, 5000000), ((7777.77, 1024000), 3, 1 << 22),
                                  ((0.0, 48000), 0, 300000)):
        segs = [(n, shift)]
        want, _ = oracle_counters(orc, segs, rate, sn0)
        for opts in (dict(), dict(walk_flags=1), dict(walk_span=3), dict(walk_waves=5)):
            for pair in PAIRS:
                c, w = doppler_amd.plan_simulate(segs, rate, sn0, 128, 2, 5, options=opts, pair=pair)
                assert (w == 1).all() and np.array_equal(c, want), (shift, opts, pair, np.flatnonzero(w != 1)[:5])


def test_header_is_plain_c(tmp_path):
    """include/doppler_hip.h is the drop-in boundary: it (and the two headers beside it) must compile as C99 on its own (no
    C++, no HIP, no torch types), each header alone as well, and a C program must link against the library with nothing else."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "doppler_hip.h"\n#include "doppler_hip_host.h"\n#include "doppler_hip_debug.h"\n#include <stdio.h>\n'
                   'int main(void) { dpx_layout l; dpx_segment s = {2048, 5000.0f}; dpx_stretch st[4]; size_t n = 0; uint32_t fin = 0;\n'
                   '  if (dpx_plan_describe(&s, 1, 1024000, 0, 3, st, 4, &n, &fin) != DPX_OK) return 2;\n'
                   '  if (dpx_plan_layout(&s, 1, 1024000, 0, 0, 0, 3, NULL, &l) != DPX_OK) return 3;\n'
                   '  printf("%d %u %llu\\n", dpx_abi_version(), fin, (unsigned long long)l.n_samples); return 0; }\n')
    exe = tmp_path / "hdr"
    lib = os.path.join(root, "doppler_amd", "lib")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
                        "-o", str(exe), "-L", lib, "-ldoppler_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)       # host-only entry points: runs without a GPU
    assert r.returncode == 0 and r.stdout.split()[1:] == ["1024", "2048"], (r.returncode, r.stdout, r.stderr[-500:])
    for h in _lib.HEADERS:                                                # every header stands alone
        one = tmp_path / ("only_" + h.replace(".h", ".c"))
        one.write_text('#include "%s"\nint main(void) { return dpx_abi_version() == DPX_ABI_VERSION ? 0 : 1; }\n' % h)
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(one)],
                           capture_output=True, text=True)
        assert r.returncode == 0, (h, r.stderr[-2000:])


def test_planner_fuzz_under_sanitizers():
    """tests/cpp/test_planner_fuzz.cpp: random plans through plan_append / finalize (all kernel choices) / simulate with
    the planner compiled under AddressSanitizer + UBSan — the hint and sentinel scans index vectors by hand, and an
    out-of-bounds read there only shows as a rare crash of the `doppler` command otherwise."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", root, "tests/cpp/test_planner_fuzz"], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr + r.stdout):
        pytest.skip("no sanitizer runtime for g++ here")
    assert r.returncode == 0, r.stderr[-1500:]
    r = subprocess.run([os.path.join(root, "tests", "cpp", "test_planner_fuzz"), "250"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ok" in r.stdout


def test_closed_form_first_reset_equals_the_candidate_scan():
    """tests/cpp/test_find_reset.cpp: dpx::find_reset (round 6: an Euclid-like descent per binade of the product — and of the
    counter, from 2^24 on — no candidate tried) against dpx::find_reset_scan (every candidate tried with the arithmetic of
    dsp.rs:125-130) on ~286 000 queries —
    named ratios from every start, random ratios of every exponent and sign, exact ties and their neighbours, dyadic
    ratios, subnormals, overflow, zero / inf / nan, small ratios whose first reset lies beyond 2^24 from starts up to 2^32."""
    import subprocess
    r = subprocess.run(["make", "-C", ROOT, "tests/cpp/test_find_reset"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_find_reset"), "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "equal to the scan" in r.stdout
    # and once more under AddressSanitizer + UBSan: the descent shifts by computed amounts and multiplies into 128 bits
    exe = os.path.join(ROOT, "tests", "cpp", "test_find_reset_san")
    csrc = os.path.join(ROOT, "doppler_amd", "csrc")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off",
                        "-fno-fast-math", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_find_reset.cpp"),
                        os.path.join(csrc, "dpx_planner.cpp"), os.path.join(csrc, "dpx_simulate.cpp")], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr + r.stdout):
        return                                         # no sanitizer runtime for g++ here: the plain run above stands
    assert r.returncode == 0, r.stderr[-1500:]
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "equal to the scan" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_chunk_sharding_seeds(orc):
    n = 2048 * 37 + 555
    for world in (1, 2, 3, 8):
        pos = 0
        for r in range(world):
            lo, hi = shard.chunk_bounds(n, world, r)
            assert lo == pos and lo % 2048 == 0
            pos = hi
            assert shard.chunk_seed(9876.543, 1024000, lo) == orc.advance_samplenum(0, 9876.543, 1024000, lo)
        assert pos == n
    segs = [(2048 * 3, 100.0), (2048 * 5, -7.5), (2048 * 2 + 9, 5000.0)]
    before, inside = shard.segments_for_chunk(segs, 2048 * 4, 2048 * 9)
    assert before == [(2048 * 3, 100.0), (2048, -7.5)] and inside == [(2048 * 4, -7.5), (2048, 5000.0)]
    sn = 0
    for cnt, hz in before:
        sn = orc.advance_samplenum(sn, hz, 1024000, cnt)
    assert shard.seed_for_segments(before, 1024000) == sn


def test_counter_wraps_like_u32(orc):
    """`*samplenum += 1` on a u32 wraps to 0 (release-build semantics, stated in oracle/doppler_oracle.c); 0 then
    resets to 1.  A ratio that never produces an integer near 2^32 keeps the counter climbing to the wrap."""
    shift, rate = 0.3, 1000003          # ratio ~3e-7: ratio*n ~ 1288.49 near n = 2^32, never an integer there
    sn0 = (1 << 32) - 150
    want, sn_end = oracle_counters(orc, [(400, shift)], rate, sn0)
    assert want[149] == (1 << 32) - 1 and want[150] == 0 and want[151] == 1
    for variant in (3, 4, 1):
        c, w = doppler_amd.plan_simulate([(400, shift)], rate, sn0, 256, 1, variant)
        assert (w == 1).all() and np.array_equal(c, want)
    st, fin = doppler_amd.plan_describe([(400, shift)], rate, sn0)
    assert fin == sn_end


def test_degenerate_ratios(orc):
    """samplerate 0 (ratio inf/NaN: never resets, counter climbs), huge shifts, negative zero."""
    for shift, rate in [(5000.0, 0), (0.0, 0), (3.0e38, 1), (-0.0, 1024000), (1e-30, 1)]:
        want, sn_end = oracle_counters(orc, [(5000, shift)], rate, 0)
        c, w = doppler_amd.plan_simulate([(5000, shift)], rate, 0, 256, 1, 3)
        assert (w == 1).all() and np.array_equal(c, want), (shift, rate)
        assert doppler_amd.plan_describe([(5000, shift)], rate, 0)[1] == sn_end


def test_random_plans_against_sequential_rule(orc):
    """Seeded random sweep: arbitrary f32 shifts (as track mode produces), rates, counter starts, multi-segment
    plans — the launch lists must reproduce the sequential counter sample for sample."""
    rng = np.random.default_rng(2024)
    for case in range(120):
        rate = int(rng.choice([8000, 48000, 256000, 1024000, 2400000, 300000, 1000003]))
        nseg = int(rng.integers(1, 5))
        segs = []
        for _ in range(nseg):
            kind = rng.integers(0, 4)
            if kind == 0:
                hz = float(np.float32(rng.integers(-20000, 20000)))
            elif kind == 1:
                hz = float(np.float32(rng.uniform(-12000, 12000)))
            elif kind == 2:
                hz = float(np.float32(rate / float(rng.choice([2, 3, 4, 5, 8, 10, 16, 100, 1000]))))
            else:
                hz = float(np.float32(rng.uniform(-3, 3)))
            segs.append((int(rng.integers(1, 9000)) if rng.random() < 0.7 else int(rng.integers(60000, 90000)), hz))
        sn0 = int(rng.choice([0, 1, 2, 1000, 65535, 1 << 20]))
        want, sn_end = oracle_counters(orc, segs, rate, sn0)
        variant = int(rng.choice([3, 4, 1, 2, 5, 5, 6]))
        block, vecs = [(256, 1), (128, 2)][case % 2]
        c, w = doppler_amd.plan_simulate(segs, rate, sn0, block, vecs, variant)
        assert writes_ok(w, segs, rate, sn0, block, vecs, variant), (case, segs, rate, sn0, variant)
        assert np.array_equal(c, want), (case, segs, rate, sn0, variant, int(np.flatnonzero(c != want)[0]))
        assert doppler_amd.plan_describe(segs, rate, sn0)[1] == sn_end


def test_every_wavefront_of_a_workgroup_reaches_its_barrier(tmp_path):
    """The kernel with a workgroup barrier (span kernel) must not let a wavefront end before it: rounds 1 and 2 did
    (s_barrier only counts live wavefronts on gfx950 — hardware behaviour, not a language guarantee).  Pinned twice:
    (1) in the source, no `return` stands between the entry of span_body and its __syncthreads(); (2) in the shipped code
    object (disassembled with llvm-objdump), every span_kernel instantiation holds exactly one s_barrier —
    if a compiler change duplicates, drops or moves barriers into divergent paths, this fails here instead of hanging
    on the GPU; kernels without LDS sharing hold none."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "doppler_amd", "csrc", "dpx_kernels.hip")).read()
    for fn in ("span_body",):
        m = re.search(r"__device__ __forceinline__ void %s\(" % fn, src)
        assert m, fn
        body = src[m.end():]
        upto = body.index("__syncthreads();")
        code = re.sub(r"//[^\n]*", "", body[:upto])
        assert not re.search(r"\breturn\b", code), "%s: a wavefront may leave before the workgroup barrier" % fn
        assert body.count("__syncthreads();", 0, body.index("\n}\n")) == 1, fn
    assert "#error" in src and "__gfx950__" in src
    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.path.join(root, "doppler_amd", "lib", "libdoppler_hip.so")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, str(tmp_path / "unused.so")])
    subprocess.check_call([llvm + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                           "--output=" + co, "--unbundle"])
    dis = subprocess.run([llvm + "/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
    counts, name = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            counts[name] = 0
        elif name and "s_barrier" in line:
            counts[name] += 1
    kern = {k: v for k, v in counts.items() if k.startswith("_ZN3dpx")}
    span = {k: v for k, v in kern.items() if "span_kernel" in k}
    assert not any("walk_kernel" in k for k in kern)                        # round 2's kernel is gone from the library
    uni = {k: v for k, v in span.items() if re.search(r"Lb1EEEv", k)}        # ..., bool UNI = true>
    multi = {k: v for k, v in span.items() if k not in uni}
    assert len(uni) == 24 and len(multi) == 32, (len(uni), len(multi))      # 4 format pairs x 2 libm builds x 3 / 4 workgroup sizes
    assert all(v == 1 for v in span.values()), {k: v for k, v in span.items() if v != 1}
    resident = {k: v for k, v in kern.items() if "resident_block_kernel" in k}     # polls a doorbell: barriers around its shared words, all uniform
    assert len(resident) == 8
    assert all(v == 0 for k, v in kern.items() if k not in span and k not in resident), {k[:60]: v for k, v in kern.items() if v and k not in span and k not in resident}


def test_resident_control_word_proves_itself(tmp_path):
    """The doorbell word of the resident block kernel (csrc/dpx_types.h, BlockCtl: ticket | payload with the ticket's tag |
    - | ticket) is read by the GPU as one 16-byte PCIe read that nothing guarantees to be atomic: every mix of dwords from
    two successive writes of a slot must fail ctl_word_valid unless it IS one of the two writes (tests/cpp/test_ctl_word.cpp
    enumerates a million of them, around the wraps of the tag and of the ticket).  Host only."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "test_ctl_word"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", str(exe), os.path.join(root, "tests", "cpp", "test_ctl_word.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "0 mixed words accepted" in r.stdout, r.stdout + r.stderr
