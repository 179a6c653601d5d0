// dpx_operators.cpp — the host-pointer operators of doppler::dsp, and the host-only arithmetic of their callers
// (one of the translation units behind include/doppler_hip*.h: see dpx_internal.h)
#include <algorithm>
#include <new>
#include <string>

#include "dpx_internal.h"
#include "host/orbit.h"
#include "host/schedule.h"

namespace dpx_api {

namespace {

// One 8 KiB block per call is what the reference's loop does (main.rs:62-99).  For such calls the fixed costs decide:
// no device staging, no hipMemcpy calls, no corrector tables (2048 samples do not pay for a table build) — the tile
// kernel reads the samples and the two plan tables from pinned host memory and writes the result there.
int run_host_small(dpx_ctx *ctx, const void *in, size_t n, int in_fmt, void *out, int out_fmt,
                   uint32_t *samplenum, float shift_hz, uint32_t samplerate)
{
    if (!ctx->small_host) {
        void *h = nullptr, *d = nullptr;
        DPX_HIP(hipHostMalloc(&h, 2 * kSmallCallBytes + kSmallPlanBytes, hipHostMallocMapped));
        hipError_t e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(h);
            return fail(DPX_ERR_HIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
        }
        ctx->small_host = static_cast<char *>(h);
        ctx->small_dev = static_cast<char *>(d);
    }
    dpx::PlanResult plan;
    uint32_t sn = *samplenum;
    dpx::plan_append(plan, dpx::ratio_of(shift_hz, samplerate), n, sn, 1 /* sincos per sample */, &ctx->periods);
    const dpx::LaunchGeom g = geometry(ctx);
    dpx::finalize(plan, g.tile(), dpx::kChooseTileOnly);
    if (plan.error) return fail(DPX_ERR_PLAN, "%s", plan.error);
    const size_t seg_bytes = align256(plan.segs.size() * sizeof(dpx::DevSeg));
    const size_t hint_bytes = plan.hint.size() * sizeof(uint32_t);
    if (plan.lut_entries != 0 || seg_bytes + hint_bytes > kSmallPlanBytes) return 1;   // caller takes the general path
    const size_t in_bytes = n * bytes_per_sample(in_fmt), out_bytes = n * bytes_per_sample(out_fmt);
    memcpy(ctx->small_host + kSmallInOff, in, in_bytes);
    memcpy(ctx->small_host + kSmallPlanOff, plan.segs.data(), plan.segs.size() * sizeof(dpx::DevSeg));
    memcpy(ctx->small_host + kSmallPlanOff + seg_bytes, plan.hint.data(), hint_bytes);
    DevPlan dev;
    dev.segs = reinterpret_cast<dpx::DevSeg *>(ctx->small_dev + kSmallPlanOff);
    dev.hint = reinterpret_cast<uint32_t *>(ctx->small_dev + kSmallPlanOff + seg_bytes);
    dev.lut = ctx->small_dev + kSmallPlanOff;      // never read: no tabulated stretch in this plan
    int rc = run_plan(plan, dev, ctx->small_dev + kSmallInOff, in_fmt, ctx->small_dev + kSmallOutOff, out_fmt, ctx->fma, g,
                      ctx->stream);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->small_host + kSmallOutOff, out_bytes);
    *samplenum = sn;
    return DPX_OK;
}

}  // namespace

// shared body of the host-pointer operators: stage in, one fused launch, stage out
int run_host(dpx_ctx *ctx, const void *in, size_t n, int in_fmt, void *out, int out_fmt,
             uint32_t *samplenum, float shift_hz, uint32_t samplerate)
{
    DPX_ENTER(ctx);
    if (n != 0 && n * 8 <= kSmallCallBytes && ctx->variant == 0) {
        const int rc = run_host_small(ctx, in, n, in_fmt, out, out_fmt, samplenum, shift_hz, samplerate);
        if (rc <= 0) return rc;
    }
    dpx::PlanResult plan;
    uint32_t sn = *samplenum;
    dpx::plan_append(plan, dpx::ratio_of(shift_hz, samplerate), n, sn, ctx->variant, &ctx->periods);
    if (n == 0) {
        *samplenum = sn;
        return DPX_OK;
    }
    const dpx::LaunchGeom g = geometry(ctx);
    dpx::finalize(plan, g.tile(), ctx->choice, ctx->tuning);
    if (plan.error) return fail(DPX_ERR_PLAN, "%s", plan.error);
    const size_t in_bytes = n * bytes_per_sample(in_fmt), out_bytes = n * bytes_per_sample(out_fmt);
    int rc = ensure_stage(ctx, in_bytes, out_bytes);
    if (rc != DPX_OK) return rc;
    if (!ctx->scratch) ctx->scratch = new (std::nothrow) DevPlan;
    if (!ctx->scratch) return fail(DPX_ERR_ARG, "out of host memory");
    // the previous call synchronised the stream, so the scratch image is free to overwrite
    rc = materialize(ctx, plan, *ctx->scratch, ctx->fma, ctx->stream);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(ctx->stage_in, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = run_plan(plan, *ctx->scratch, ctx->stage_in, in_fmt, ctx->stage_out, out_fmt, ctx->fma, g, ctx->stream);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(out, ctx->stage_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    DPX_HIP(hipStreamSynchronize(ctx->stream));
    *samplenum = sn;
    return DPX_OK;
}

// the same for a list of constant-shift segments (host pointers): one plan, one fused launch
int run_host_segments(dpx_ctx *ctx, const void *in, int in_fmt, void *out, int out_fmt, uint32_t *samplenum,
                      const dpx_segment *segs, size_t n_segs, uint32_t samplerate)
{
    DPX_ENTER(ctx);
    dpx::PlanResult plan;
    uint32_t sn = *samplenum;
    append_segments(plan, segs, n_segs, samplerate, sn, ctx->variant, ctx->periods);
    const uint64_t n = plan.n_samples;
    if (n == 0) {
        *samplenum = sn;
        return DPX_OK;
    }
    const dpx::LaunchGeom g = geometry(ctx);
    dpx::finalize(plan, g.tile(), ctx->choice, ctx->tuning);
    if (plan.error) return fail(DPX_ERR_PLAN, "%s", plan.error);
    const size_t in_bytes = n * bytes_per_sample(in_fmt), out_bytes = n * bytes_per_sample(out_fmt);
    int rc = ensure_stage(ctx, in_bytes, out_bytes);
    if (rc != DPX_OK) return rc;
    if (!ctx->scratch) ctx->scratch = new (std::nothrow) DevPlan;
    if (!ctx->scratch) return fail(DPX_ERR_ARG, "out of host memory");
    rc = materialize(ctx, plan, *ctx->scratch, ctx->fma, ctx->stream);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(ctx->stage_in, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = run_plan(plan, *ctx->scratch, ctx->stage_in, in_fmt, ctx->stage_out, out_fmt, ctx->fma, g, ctx->stream);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(out, ctx->stage_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    DPX_HIP(hipStreamSynchronize(ctx->stream));
    *samplenum = sn;
    return DPX_OK;
}

}  // namespace dpx_api

using namespace dpx_api;

extern "C" {

/* ------------------------------------------------------------ host operators */

int dpx_shift_block(dpx_ctx *ctx, const void *in, size_t in_bytes, int in_fmt, void *out,
                    size_t out_cap, int out_fmt, uint32_t *samplenum, float shift_hz,
                    uint32_t samplerate, size_t *n_samples_out)
{
    if (!ctx || !samplenum || (!in && in_bytes) || !fmt_ok(in_fmt) || !fmt_ok(out_fmt))
        return fail(DPX_ERR_ARG, "bad argument");
    if (in_bytes % bytes_per_sample(in_fmt) != 0)
        return fail(DPX_ERR_BLOCK_LEN, "%zu bytes is not a whole number of %s samples", in_bytes,
                    in_fmt == DPX_FMT_I16 ? "i16" : "f32");
    const size_t n = in_bytes / bytes_per_sample(in_fmt);
    if (n * bytes_per_sample(out_fmt) > out_cap || (!out && n))
        return fail(DPX_ERR_CAPACITY, "output needs %zu bytes, capacity %zu", n * bytes_per_sample(out_fmt), out_cap);
    // a reference-sized block goes to the resident kernel when nothing is in flight there (a doorbell and a completion word
    // instead of a launch and a stream synchronisation: profiles/r04_cli.md); everything else as before
    if (ctx->resident_on && n != 0 && n * 8 <= kSmallCallBytes && ctx->variant == 0) {
        bool busy = false;
        for (const dpx_ctx::AsyncSlot &a : ctx->async_slots) busy = busy || a.seq != 0;
        if (!busy) {
            dpx_ticket t = 0;
            uint32_t sn = *samplenum;
            int rc = dpx_shift_block_async(ctx, in, in_bytes, in_fmt, out_fmt, &sn, shift_hz, samplerate, &t);
            if (rc == DPX_OK) rc = dpx_wait(ctx, t, out, out_cap, n_samples_out);
            if (rc == DPX_OK) *samplenum = sn;
            return rc;
        }
    }
    int rc = run_host(ctx, in, n, in_fmt, out, out_fmt, samplenum, shift_hz, samplerate);
    if (rc == DPX_OK && n_samples_out) *n_samples_out = n;
    return rc;
}

int dpx_shift_blocks(dpx_ctx *ctx, const void *in, size_t in_bytes, int in_fmt, void *out, size_t out_cap, int out_fmt,
                     uint32_t *samplenum, const float *shift_hz, size_t n_blocks, uint32_t samplerate, size_t *n_samples_out)
{
    if (!ctx || !samplenum || (!in && in_bytes) || !fmt_ok(in_fmt) || !fmt_ok(out_fmt) || (n_blocks && !shift_hz))
        return fail(DPX_ERR_ARG, "bad argument");
    const size_t ibs = bytes_per_sample(in_fmt);
    if (n_blocks != (in_bytes + DPX_BUFFER_SIZE - 1) / DPX_BUFFER_SIZE)
        return fail(DPX_ERR_ARG, "%zu bytes are %zu blocks of %d bytes, not %zu", in_bytes, (in_bytes + DPX_BUFFER_SIZE - 1) / DPX_BUFFER_SIZE,
                    DPX_BUFFER_SIZE, n_blocks);
    if (in_bytes % ibs != 0)
        return fail(DPX_ERR_BLOCK_LEN, "%zu bytes is not a whole number of %s samples", in_bytes, in_fmt == DPX_FMT_I16 ? "i16" : "f32");
    const size_t n = in_bytes / ibs;
    if (n * bytes_per_sample(out_fmt) > out_cap || (!out && n))
        return fail(DPX_ERR_CAPACITY, "output needs %zu bytes, capacity %zu", n * bytes_per_sample(out_fmt), out_cap);
    // runs of blocks with the same shift (bit pattern) become one segment: same arithmetic, fewer stretches
    std::vector<dpx_segment> segs;
    const size_t spb = DPX_BUFFER_SIZE / ibs;
    for (size_t b = 0; b < n_blocks; ++b) {
        const uint64_t cnt = std::min<uint64_t>(spb, n - b * spb);
        if (!segs.empty() && memcmp(&segs.back().shift_hz, &shift_hz[b], sizeof(float)) == 0) segs.back().n_samples += cnt;
        else segs.push_back({cnt, shift_hz[b]});
    }
    int rc = run_host_segments(ctx, in, in_fmt, out, out_fmt, samplenum, segs.data(), segs.size(), samplerate);
    if (rc == DPX_OK && n_samples_out) *n_samples_out = n;
    return rc;
}

int dpx_shift_frequency(dpx_ctx *ctx, const dpx_complex32 *inbuf, size_t n, uint32_t *samplenum,
                        float shift_hz, uint32_t samplerate, dpx_complex32 *out)
{
    if (!ctx || !samplenum || (n && (!inbuf || !out))) return fail(DPX_ERR_ARG, "bad argument");
    // Complex<f32> in memory is exactly the f32 wire format (dsp.rs:108-109, main.rs:91)
    return run_host(ctx, inbuf, n, DPX_FMT_F32, out, DPX_FMT_F32, samplenum, shift_hz, samplerate);
}

int dpx_convert_iqi16_to_complex(dpx_ctx *ctx, const uint8_t *inbuf, size_t in_bytes,
                                 dpx_complex32 *out, size_t out_cap, size_t *n_out)
{
    if (!ctx || (in_bytes && (!inbuf || !out))) return fail(DPX_ERR_ARG, "bad argument");
    if (in_bytes % 4 != 0) return fail(DPX_ERR_BLOCK_LEN, "assertion failed: inbuf.len() %% 4 == 0");
    const size_t n = in_bytes / 4;
    if (n > out_cap) return fail(DPX_ERR_CAPACITY, "output needs %zu samples, capacity %zu", n, out_cap);
    if (n_out) *n_out = n;
    if (n == 0) return DPX_OK;
    DPX_ENTER(ctx);
    int rc = ensure_stage(ctx, in_bytes, n * 8);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(ctx->stage_in, inbuf, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = dpx::launch_unpack_i16(ctx->stage_in, ctx->stage_out, n, ctx->stream);
    if (rc != DPX_OK) return fail(rc, "kernel launch failed");
    DPX_HIP(hipMemcpyAsync(out, ctx->stage_out, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    DPX_HIP(hipStreamSynchronize(ctx->stream));
    return DPX_OK;
}

int dpx_convert_iqf32_to_complex(dpx_ctx *ctx, const uint8_t *inbuf, size_t in_bytes,
                                 dpx_complex32 *out, size_t out_cap, size_t *n_out)
{
    if (!ctx || (in_bytes && (!inbuf || !out))) return fail(DPX_ERR_ARG, "bad argument");
    if (in_bytes % 8 != 0) return fail(DPX_ERR_BLOCK_LEN, "assertion failed: inbuf.len() %% 8 == 0");
    const size_t n = in_bytes / 8;
    if (n > out_cap) return fail(DPX_ERR_CAPACITY, "output needs %zu samples, capacity %zu", n, out_cap);
    if (n_out) *n_out = n;
    // dsp.rs:108-109 is a bit-for-bit reinterpretation: no arithmetic, no device work
    if (n) memcpy(out, inbuf, in_bytes);
    return DPX_OK;
}

int dpx_pack_iqi16(dpx_ctx *ctx, const dpx_complex32 *inbuf, size_t n, uint8_t *out, size_t out_cap)
{
    if (!ctx || (n && (!inbuf || !out))) return fail(DPX_ERR_ARG, "bad argument");
    if (n * 4 > out_cap) return fail(DPX_ERR_CAPACITY, "output needs %zu bytes, capacity %zu", n * 4, out_cap);
    if (n == 0) return DPX_OK;
    DPX_ENTER(ctx);
    int rc = ensure_stage(ctx, n * 8, n * 4);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(ctx->stage_in, inbuf, n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = dpx::launch_pack_i16(ctx->stage_in, ctx->stage_out, n, ctx->stream, ctx->i16_cast == DPX_CAST_LEGACY_X86);
    if (rc != DPX_OK) return fail(rc, "kernel launch failed");
    DPX_HIP(hipMemcpyAsync(out, ctx->stage_out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    DPX_HIP(hipStreamSynchronize(ctx->stream));
    return DPX_OK;
}

int dpx_ccexpf(dpx_ctx *ctx, dpx_complex32 *z, size_t n)
{
    if (!ctx || (n && !z)) return fail(DPX_ERR_ARG, "bad argument");
    if (n == 0) return DPX_OK;
    DPX_ENTER(ctx);
    int rc = ensure_stage(ctx, n * 8, 0);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(ctx->stage_in, z, n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = dpx::launch_ccexpf(ctx->stage_in, n, ctx->fma, ctx->stream);
    if (rc != DPX_OK) return fail(rc, "kernel launch failed");
    DPX_HIP(hipMemcpyAsync(z, ctx->stage_in, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    DPX_HIP(hipStreamSynchronize(ctx->stream));
    return DPX_OK;
}

int dpx_ccexpf_imag(dpx_ctx *ctx, dpx_complex32 *z, size_t n)
{
    if (!ctx || (n && !z)) return fail(DPX_ERR_ARG, "bad argument");
    if (n == 0) return DPX_OK;
    DPX_ENTER(ctx);
    int rc = ensure_stage(ctx, n * 8, 0);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(ctx->stage_in, z, n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = dpx::launch_ccexpf_imag(ctx->stage_in, n, ctx->fma, ctx->stream);
    if (rc != DPX_OK) return fail(rc, "kernel launch failed");
    DPX_HIP(hipMemcpyAsync(z, ctx->stage_in, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    DPX_HIP(hipStreamSynchronize(ctx->stream));
    return DPX_OK;
}

/* ------------------------------------------------------------ counter algebra */

int dpx_find_reset(float shift_hz, uint32_t samplerate, uint32_t n_start, uint64_t max_scan,
                   uint32_t *n_reset, int *found)
{
    if (!n_reset || !found) return fail(DPX_ERR_ARG, "bad argument");
    uint32_t n1 = 0;
    *found = dpx::find_reset(dpx::ratio_of(shift_hz, samplerate), n_start, max_scan, &n1) ? 1 : 0;
    *n_reset = n1;
    return DPX_OK;
}

int dpx_find_reset_scan(float shift_hz, uint32_t samplerate, uint32_t n_start, uint64_t max_scan,
                        uint32_t *n_reset, int *found)
{
    if (!n_reset || !found) return fail(DPX_ERR_ARG, "bad argument");
    uint32_t n1 = 0;
    *found = dpx::find_reset_scan(dpx::ratio_of(shift_hz, samplerate), n_start, max_scan, &n1) ? 1 : 0;
    *n_reset = n1;
    return DPX_OK;
}

int dpx_samplenum_after_segments(const dpx_segment *segs, size_t n_segs, uint32_t samplerate, uint32_t samplenum0,
                                 uint32_t *samplenum)
{
    if (!samplenum || (n_segs && !segs)) return fail(DPX_ERR_ARG, "bad argument");
    uint32_t sn = samplenum0;
    dpx::PeriodCache cache;
    for (size_t i = 0; i < n_segs; ++i) {
        dpx::PlanResult plan;                          // (the stretch list of one segment: a handful of entries, dropped at once)
        dpx::plan_append(plan, dpx::ratio_of(segs[i].shift_hz, samplerate), segs[i].n_samples, sn, 1, &cache);
    }
    *samplenum = sn;
    return DPX_OK;
}

int dpx_samplenum_after(float shift_hz, uint32_t samplerate, uint32_t samplenum0, uint64_t k,
                        uint32_t *samplenum)
{
    if (!samplenum) return fail(DPX_ERR_ARG, "bad argument");
    dpx::PlanResult plan;
    uint32_t sn = samplenum0;
    dpx::plan_append(plan, dpx::ratio_of(shift_hz, samplerate), k, sn, 1);
    *samplenum = sn;
    return DPX_OK;
}

int dpx_track_schedule(const double *range_rate_km_s, size_t n_table, uint32_t samplerate,
                       uint32_t frequency_hz, int32_t offset_hz, int has_offset, int in_fmt,
                       uint64_t in_bytes, float *shift_hz, size_t cap, size_t *n_blocks)
{
    if (!range_rate_km_s || n_table == 0 || !n_blocks || (cap && !shift_hz) || !fmt_ok(in_fmt))
        return fail(DPX_ERR_ARG, "bad argument");
    dpx::ReplaySchedule sch(
        [=](int64_t dt) {
            const size_t i = dt < 0 ? 0 : ((uint64_t)dt >= n_table ? n_table - 1 : (size_t)dt);
            return range_rate_km_s[i];
        },
        samplerate, frequency_hz, has_offset != 0, offset_hz);
    const size_t bps = bytes_per_sample(in_fmt);
    uint64_t pos = 0;
    size_t nb = 0;
    for (;;) {
        const uint64_t take = in_bytes - pos < DPX_BUFFER_SIZE ? in_bytes - pos : DPX_BUFFER_SIZE;
        const float hz = sch.next_block_shift();
        if (nb < cap) shift_hz[nb] = hz;
        ++nb;
        if (take % bps != 0) return fail(DPX_ERR_BLOCK_LEN, "trailing partial sample");
        pos += take;
        if (take != DPX_BUFFER_SIZE) break;
        sch.block_done((size_t)(take / bps));
    }
    *n_blocks = nb;
    return DPX_OK;
}

int dpx_orbit_observe(const char *l1, const char *l2, double lat_deg, double lon_deg, double alt_m,
                      double unix_time_s, double out[4])
{
    if (!l1 || !l2 || !out) return fail(DPX_ERR_ARG, "bad argument");
    dpx::Tle tle;
    std::string err;
    if (!dpx::tle_parse(l1, l2, &tle, &err)) return fail(DPX_ERR_ARG, "%s", err.c_str());
    dpx::Sgp4 prop;
    if (!prop.init(tle, &err)) return fail(DPX_ERR_ARG, "%s", err.c_str());
    dpx::Observer obs;
    obs.lat_deg = lat_deg;
    obs.lon_deg = lon_deg;
    obs.alt_m = alt_m;
    const dpx::LookAngles la = prop.observe(obs, unix_time_s);
    out[0] = la.az_deg;
    out[1] = la.el_deg;
    out[2] = la.range_km;
    out[3] = la.range_rate_km_s;
    return DPX_OK;
}

int dpx_orbit_propagate(const char *l1, const char *l2, double tsince_min, double out[6])
{
    if (!l1 || !l2 || !out) return fail(DPX_ERR_ARG, "bad argument");
    dpx::Tle tle;
    std::string err;
    if (!dpx::tle_parse(l1, l2, &tle, &err)) return fail(DPX_ERR_ARG, "%s", err.c_str());
    dpx::Sgp4 prop;
    if (!prop.init(tle, &err)) return fail(DPX_ERR_ARG, "%s", err.c_str());
    prop.propagate(tsince_min, out, out + 3);
    return DPX_OK;
}

}  // extern "C"
