"""The slab ring (dpx_stream_*, reference src/main.rs:57-99 gathered into slabs) on each of its PCIe paths against the
oracle: DIRECT (the fused kernel loads from / stores to the pinned host slabs — the default), STAGED (copy engines either
side of an HBM-to-HBM launch), and the two mixed forms.  Every comparison is exact equality of output bytes."""
import numpy as np
import pytest

from helpers import BPS, assert_same_bytes, make_iq

pytestmark = pytest.mark.gpu

PATHS = ["direct", "staged", "direct_in", "direct_out", "staged_per_slab"]
NONCOHERENT, NUMAUSER = 0x80000000, 0x20000000


def drive(st, x, bi, slabs, n_ring):
    """slabs: [(n_samples, segments)]; returns the concatenated output, feeding the ring to capacity."""
    outs, pos = [], 0
    for n, segs in slabs:
        if st.pending() == n_ring:
            outs.append(st.next())
        buf = st.acquire()
        buf[: n * bi] = x[pos * bi:(pos + n) * bi]
        st.submit(n * bi, segs)
        pos += n
    while st.pending():
        outs.append(st.next())
    return np.concatenate(outs) if outs else np.empty(0, np.uint8)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("intype,outtype,shift,rate", [("i16", "i16", 5000, 1024000), ("f32", "f32", -15000, 256000),
                                                       ("i16", "f32", 5001, 1024000), ("f32", "i16", 3, 1024000)])
def test_const_stream_through_the_ring(ctx, orc, path, intype, outtype, shift, rate):
    """`doppler const`: slabs of full and ragged sizes, every slab buffer used several times with new content (a kernel that
    reads host memory must never see a previous lap's bytes), counter carried — equal to the oracle's sequential pass over the whole stream."""
    import doppler_amd
    bi = BPS[intype]
    slab_bytes, n_ring = 1 << 19, 3
    per = slab_bytes // bi
    sizes = [per, per, per - 8, per, 1, per, per // 2 + 3, per, per, 2048, per, per - 1]
    n = sum(sizes)
    x = make_iq(intype, n, 1234, full_scale=True)
    st = doppler_amd.Stream(ctx, intype, outtype, rate, slab_bytes=slab_bytes, n_slabs=n_ring, path=path)
    try:
        assert st.describe()["path"] == path and not st.describe()["copy_only"]
        got = drive(st, x, bi, [(m, [(m, float(shift))]) for m in sizes], n_ring)
        want, sn = orc.segments_stream(x, intype, outtype, [(n, float(shift))], rate, threads=8)
        assert st.samplenum == sn
        assert_same_bytes(got, want, outtype, "ring path %s" % path)
    finally:
        st.close()


@pytest.mark.parametrize("path", PATHS)
def test_track_segments_through_the_ring(ctx, orc, path):
    """`doppler track`: shift changes inside and between slabs (span and tile kernels reading / writing host memory)."""
    import doppler_amd
    rate = 1024000
    rng = np.random.default_rng(5)
    slab_bytes, n_ring = 1 << 20, 4
    per = slab_bytes // 4
    slabs, all_segs = [], []
    for k in range(11):
        m = per if k % 3 else int(rng.integers(1, per))
        cuts = sorted(set(int(c) for c in rng.integers(1, m, size=3))) if m > 4 else []
        bounds = [0] + cuts + [m]
        segs = [(b - a, float(np.float32(4000.0 + 13.25 * k + i))) for i, (a, b) in enumerate(zip(bounds[:-1], bounds[1:]))]
        slabs.append((m, segs))
        all_segs += segs
    n = sum(m for m, _ in slabs)
    x = make_iq("i16", n, 77, full_scale=True)
    st = doppler_amd.Stream(ctx, "i16", "i16", rate, slab_bytes=slab_bytes, n_slabs=n_ring, path=path)
    try:
        got = drive(st, x, 4, slabs, n_ring)
        want, sn = orc.segments_stream(x, "i16", "i16", all_segs, rate, threads=8)
        assert st.samplenum == sn
        assert_same_bytes(got, want, "i16", "track ring path %s" % path)
    finally:
        st.close()


def test_default_path_follows_the_slab_size_and_the_environment_selects(ctx, orc, monkeypatch):
    """Small slabs (a live pipe's 8 KiB block) take the one-launch direct path, large ones the paced copy engines; in between
    engine H2D against the kernel's own stores (dpx_stream.cpp: kDirectBelow, kStagedFrom)."""
    import doppler_amd
    for slab_bytes, want in ((8192, "direct"), (1 << 16, "direct"), (1 << 20, "direct_out"), (2 << 20, "direct_out"),
                             (4 << 20, "staged"), (16 << 20, "staged")):
        st = doppler_amd.Stream(ctx, "i16", "i16", 1024000, slab_bytes=slab_bytes, n_slabs=2)
        d = st.describe()
        assert d["path"] == want and not d["unpaced"], (slab_bytes, d)
        if want == "staged":
            # the ring has made sure its three streams do not wait for each other (separate_lane_streams)
            assert d["probe_rounds"] >= 1 and not d["streams_share_a_queue"], d
        st.close()
    st = doppler_amd.Stream(ctx, "i16", "i16", 1024000, slab_bytes=1 << 16, n_slabs=2)
    monkeypatch.setenv("DPX_STREAM_PATH", "2")
    st = doppler_amd.Stream(ctx, "i16", "i16", 1024000, slab_bytes=1 << 16, n_slabs=2)
    assert st.describe()["path"] == "staged"
    st.close()
    monkeypatch.setenv("DPX_STREAM_PATH", "9")
    with pytest.raises(doppler_amd.DspError):
        doppler_amd.Stream(ctx, "i16", "i16", 1024000, slab_bytes=1 << 16, n_slabs=2)


@pytest.mark.parametrize("kw", [dict(unpaced=True), dict(no_probe=True), dict(unpaced=True, no_probe=True)])
def test_staged_path_without_pacing_or_probe_keeps_the_bytes(ctx, orc, kw):
    """The A/B switches of the staged path (every D2H queued at submit time; streams as the runtime deals them) change the rate only."""
    import doppler_amd
    rate, shift = 1024000, 5001
    slab_bytes, n_ring = 4 << 20, 5
    per = slab_bytes // 4
    sizes = [per] * 7 + [per - 12, 3, per]
    x = make_iq("i16", sum(sizes), 55, full_scale=True)
    st = doppler_amd.Stream(ctx, "i16", "i16", rate, slab_bytes=slab_bytes, n_slabs=n_ring, path="staged", **kw)
    try:
        got = drive(st, x, 4, [(m, [(m, float(shift))]) for m in sizes], n_ring)
        want, sn = orc.segments_stream(x, "i16", "i16", [(sum(sizes), float(shift))], rate, threads=8)
        assert st.samplenum == sn
        assert_same_bytes(got, want, "i16", "staged ring %r" % (kw,))
    finally:
        st.close()


@pytest.mark.parametrize("flags", [(NONCOHERENT, NONCOHERENT), (NONCOHERENT, 0), (0, NONCOHERENT), (NUMAUSER, NUMAUSER)])
def test_direct_path_with_other_host_memory_kinds(ctx, orc, flags):
    """The A/B's alternatives keep the bytes: non-coherent (GPU-cacheable) host slabs rewritten by the CPU between laps."""
    import doppler_amd
    rate, shift = 1024000, 5000
    slab_bytes, n_ring = 1 << 18, 2
    per = slab_bytes // 4
    sizes = [per] * 9 + [per - 4]
    x = make_iq("i16", sum(sizes), 4321)
    st = doppler_amd.Stream(ctx, "i16", "i16", rate, slab_bytes=slab_bytes, n_slabs=n_ring, path="direct",
                            in_host_flags=flags[0], out_host_flags=flags[1])
    try:
        got = drive(st, x, 4, [(m, [(m, float(shift))]) for m in sizes], n_ring)
        want, sn = orc.segments_stream(x, "i16", "i16", [(sum(sizes), float(shift))], rate, threads=8)
        assert st.samplenum == sn
        assert_same_bytes(got, want, "i16", "direct ring, host flags %r" % (flags,))
    finally:
        st.close()


@pytest.mark.parametrize("path", PATHS)
def test_copy_only_calibration_moves_the_bytes_it_claims(ctx, path):
    """DPX_STREAM_COPY_ONLY is the `peak` of bench.py's extra.stream_ring: on the paths with a kernel side the slab's bytes must
    really cross (output == input for equal formats); the engine-only path moves both slabs' worth without relating them."""
    import doppler_amd
    slab_bytes = 1 << 18
    x = make_iq("i16", 3 * slab_bytes // 4, 9)
    st = doppler_amd.Stream(ctx, "i16", "i16", 1024000, slab_bytes=slab_bytes, n_slabs=2, path=path, copy_only=True)
    try:
        assert st.describe()["copy_only"]
        got = drive(st, x, 4, [(slab_bytes // 4, [(slab_bytes // 4, 5000.0)])] * 3, 2)
        assert got.size == x.size
        if not path.startswith("staged"):
            assert np.array_equal(got, x)
    finally:
        st.close()


def test_rccl_gather_on_one_gpu_sends_to_itself(ctx, orc):
    """gather="rccl" (BASELINE.json's north_star form: outputs over RCCL into the first GPU, from there to the host) on the one
    GPU a test box has: ncclCommInitAll over one device, every slab through an ncclSend / ncclRecv pair to itself
    (DPX_STREAM_GATHER_SELF), then the paced D2H — the machinery a ring over N GPUs runs per slab, never executed between
    two physical devices here (README.md says so).  Bytes against the oracle; a device listed twice is refused."""
    import doppler_amd
    rate = 1024000
    slab_bytes, n_ring = 1 << 20, 3
    per = slab_bytes // 4
    sizes = [per, per, per - 40, per, 7, per, per, per // 3, per, per]
    x = make_iq("i16", sum(sizes), 2024, full_scale=True)
    segs = [(m, float(np.float32(5001.0 + 3.0 * i))) for i, m in enumerate(sizes)]
    st = doppler_amd.Stream(ctx, "i16", "f32", rate, slab_bytes=slab_bytes, n_slabs=n_ring, gather="rccl", gather_self=True)
    try:
        d = st.describe()
        assert d["gather"] == "rccl" and d["path"] == "staged", d
        got = drive(st, x, 4, [(m, [sg]) for m, sg in zip(sizes, segs)], n_ring)
        want, sn = orc.segments_stream(x, "i16", "f32", segs, rate, threads=8)
        assert st.samplenum == sn
        assert_same_bytes(got, want, "f32", "RCCL gather, self send/recv")
    finally:
        st.close()
    # without the self flag a one-GPU ring has nothing to gather: the communicator is made, no slab travels
    st = doppler_amd.Stream(ctx, "i16", "i16", rate, slab_bytes=slab_bytes, n_slabs=2, gather="rccl")
    try:
        got = drive(st, x[: 3 * slab_bytes], 4, [(per, [(per, 5000.0)])] * 3, 2)
        want, _ = orc.segments_stream(x[: 3 * slab_bytes], "i16", "i16", [(3 * per, 5000.0)], rate, threads=8)
        assert_same_bytes(got, want, "i16", "RCCL gather over one GPU")
    finally:
        st.close()
    other = doppler_amd.Context(0)
    try:
        with pytest.raises(doppler_amd.DspError) as e:
            doppler_amd.Stream([ctx, other], "i16", "i16", rate, slab_bytes=slab_bytes, n_slabs=2, gather="rccl")
        assert "distinct devices" in str(e.value)
        with pytest.raises(doppler_amd.DspError):
            doppler_amd.Stream(ctx, "i16", "i16", rate, slab_bytes=slab_bytes, n_slabs=2, gather="rccl", path="direct")
    finally:
        other.close()
