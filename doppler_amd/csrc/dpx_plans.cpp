// dpx_plans.cpp — plans over device buffers (the bulk API) and the planner's self-checks
// (one of the translation units behind include/doppler_hip*.h: see dpx_internal.h)
#include <new>

#include "dpx_internal.h"

using namespace dpx_api;

extern "C" {

int dpx_plan_describe(const dpx_segment *segs, size_t n_segs, uint32_t samplerate,
                      uint32_t samplenum0, int variant, dpx_stretch *out, size_t cap,
                      size_t *n_out, uint32_t *final_samplenum)
{
    if ((n_segs && !segs) || !n_out || (cap && !out)) return fail(DPX_ERR_ARG, "bad argument");
    static_assert(sizeof(dpx_stretch) == sizeof(dpx::StretchView), "dpx_stretch mirrors the head of DevSeg");
    dpx::PlanResult plan;
    uint32_t sn = samplenum0;
    dpx::PeriodCache cache;
    append_segments(plan, segs, n_segs, samplerate, sn, variant, cache);
    *n_out = plan.segs.size();
    for (size_t i = 0; i < plan.segs.size() && i < cap; ++i) memcpy(&out[i], &plan.segs[i], sizeof(dpx_stretch));
    if (final_samplenum) *final_samplenum = sn;
    return DPX_OK;
}

int dpx_plan_simulate(const dpx_segment *segs, size_t n_segs, uint32_t samplerate,
                      uint32_t samplenum0, int block, int vecs, int variant, const dpx_options *opt,
                      int in_fmt, int out_fmt, uint32_t *counters, uint8_t *writes, uint64_t n_samples)
{
    if ((n_segs && !segs) || !counters || !writes || !fmt_ok(in_fmt) || !fmt_ok(out_fmt)) return fail(DPX_ERR_ARG, "bad argument");
    dpx::PlanResult plan;
    uint32_t sn = samplenum0;
    const int v = variant >= 3 ? 0 : variant;
    dpx::PeriodCache cache;
    append_segments(plan, segs, n_segs, samplerate, sn, v, cache);
    if (plan.n_samples != n_samples) return fail(DPX_ERR_PLAN, "segments hold %llu samples, buffers %llu",
                                                 (unsigned long long)plan.n_samples, (unsigned long long)n_samples);
    dpx::LaunchGeom g;
    g.block = block ? block : 128;
    g.vecs = vecs ? vecs : 2;
    dpx::finalize(plan, g.tile(), choice_of(variant), tuning_of(opt));
    memset(writes, 0, n_samples);
    dpx::simulate(plan, counters, writes, in_fmt, out_fmt);
    return DPX_OK;
}

int dpx_plan_layout(const dpx_segment *segs, size_t n_segs, uint32_t samplerate, uint32_t samplenum0,
                    int block, int vecs, int variant, const dpx_options *opt, dpx_layout *out)
{
    if ((n_segs && !segs) || !out) return fail(DPX_ERR_ARG, "bad argument");
    dpx::PlanResult plan;
    uint32_t sn = samplenum0;
    const int v = variant >= 3 ? 0 : variant;
    dpx::PeriodCache cache;
    append_segments(plan, segs, n_segs, samplerate, sn, v, cache);
    dpx::LaunchGeom g;
    g.block = block ? block : 128;
    g.vecs = vecs ? vecs : 2;
    dpx::finalize(plan, g.tile(), choice_of(variant), tuning_of(opt));
    if (plan.error) return fail(DPX_ERR_PLAN, "%s", plan.error);
    memset(out, 0, sizeof *out);
    out->n_samples = plan.n_samples;
    out->n_stretches = (uint32_t)plan.segs.size();
    out->table_entries = plan.lut_entries;
    out->f32_i16_by_tiles = plan.whole_tiles.empty() ? 0u : 1u;
    for (const dpx::Launch &ln : plan.launches) {
        if (ln.kind == 0) {
            ++out->rows_launches;
            out->rows_samples += ln.rows.B - ln.rows.A;
            out->single_samples += (ln.rows.A - ln.rows.r0) + (ln.rows.r1 - ln.rows.B);
        } else if (ln.kind == 1) {
            ++out->tile_launches;
            out->tile_samples += ln.tiles.m1 - ln.tiles.m0;
        } else {
            ++out->walk_launches;
            out->leftover_workgroups = ln.walk.n_left_wg;
        }
    }
    if (!plan.walk.empty()) {
        out->leftover_ranges = (uint32_t)plan.left.size() - 1;
        for (size_t i = 0; i + 1 < plan.walk.size(); ++i) {
            if (plan.walk[i].upw == 0) continue;                  // a group of leftover blocks
            out->walk_workgroups += (plan.walk[i].nwg + 7u) & ~7u; // spans are padded to multiples of 8 workgroups
            if (plan.walk[i].row0 != 0) continue;                 // one descriptor per span: count matrices once
            ++out->walk_matrices;
            out->walk_samples += plan.walk[i].E - plan.walk[i].A;
        }
        for (size_t i = 0; i + 1 < plan.left.size(); ++i) out->single_samples += plan.left[i].len;
    }
    return DPX_OK;
}

/* ------------------------------------------------------------------- bulk API */

int dpx_plan_segments(dpx_ctx *ctx, const dpx_segment *segs, size_t n_segs, uint32_t samplerate,
                      uint32_t samplenum0, dpx_plan **out)
{
    if (!ctx || !out || (n_segs && !segs)) return fail(DPX_ERR_ARG, "bad argument");
    *out = nullptr;
    dpx_plan *p = new (std::nothrow) dpx_plan;
    if (!p) return fail(DPX_ERR_ARG, "out of host memory");
    p->ctx = ctx;
    p->geom = geometry(ctx);
    p->fma = ctx->fma;
    uint32_t sn = samplenum0;
    p->host.final_samplenum = sn;
    using clk = std::chrono::steady_clock;
    auto us_since = [](clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); };
    clk::time_point t0 = clk::now();
    append_segments(p->host, segs, n_segs, samplerate, sn, ctx->variant, ctx->periods);
    p->t_append_us = us_since(t0);
    t0 = clk::now();
    dpx::finalize(p->host, p->geom.tile(), ctx->choice, ctx->tuning);
    p->t_finalize_us = us_since(t0);
    t0 = clk::now();
    // the device's lock from here to the end: no other thread's context starts a resident block kernel on this device
    // between its being asked to leave and this plan's upload having finished (the upload would queue behind it)
    std::unique_lock<std::recursive_mutex> lock(ctx->dev->mu);
    hipError_t e = hipSetDevice(ctx->device);
    int rc = e == hipSuccess ? DPX_OK : fail(DPX_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    if (rc == DPX_OK) rc = resident_stop_device(ctx);
    if (rc == DPX_OK && p->host.error) rc = fail(DPX_ERR_PLAN, "%s", p->host.error);
    if (rc == DPX_OK && p->host.n_samples) {
        rc = materialize(ctx, p->host, p->dev, p->fma, ctx->stream);
        if (rc == DPX_OK) {
            e = hipStreamSynchronize(ctx->stream);   // tables are complete before any user stream runs
            if (e != hipSuccess) rc = fail(DPX_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
        }
    }
    lock.unlock();
    p->t_upload_us = us_since(t0);
    if (rc != DPX_OK) {
        dpx_plan_destroy(p);
        return rc;
    }
    *out = p;
    return DPX_OK;
}

int dpx_plan_const(dpx_ctx *ctx, float shift_hz, uint32_t samplerate, uint32_t samplenum0,
                   uint64_t n_samples, dpx_plan **out)
{
    dpx_segment s;
    s.n_samples = n_samples;
    s.shift_hz = shift_hz;
    return dpx_plan_segments(ctx, &s, 1, samplerate, samplenum0, out);
}

int dpx_plan_n_samples(const dpx_plan *plan, uint64_t *n_samples)
{
    if (!plan || !n_samples) return fail(DPX_ERR_ARG, "bad argument");
    *n_samples = plan->host.n_samples;
    return DPX_OK;
}

int dpx_plan_timing(const dpx_plan *plan, double out_us[3])
{
    if (!plan || !out_us) return fail(DPX_ERR_ARG, "bad argument");
    out_us[0] = plan->t_append_us;
    out_us[1] = plan->t_finalize_us;
    out_us[2] = plan->t_upload_us;
    return DPX_OK;
}

int dpx_plan_final_samplenum(const dpx_plan *plan, uint32_t *samplenum)
{
    if (!plan || !samplenum) return fail(DPX_ERR_ARG, "bad argument");
    *samplenum = plan->host.final_samplenum;
    return DPX_OK;
}

void dpx_plan_destroy(dpx_plan *plan)
{
    if (!plan) return;
    if (plan->dev.buf) {
        // hipFree waits for the device: a resident block kernel (any context's) leaves first, under the device's lock
        std::lock_guard<std::recursive_mutex> lock(plan->ctx->dev->mu);
        (void)hipSetDevice(plan->ctx->device);
        (void)resident_stop_device(plan->ctx);
        release(plan->dev);
    }
    delete plan;
}

int dpx_run_device(dpx_plan *plan, const void *d_in, int in_fmt, void *d_out, int out_fmt,
                   void *hip_stream)
{
    if (!plan || !fmt_ok(in_fmt) || !fmt_ok(out_fmt)) return fail(DPX_ERR_ARG, "bad argument");
    if (plan->host.n_samples == 0) return DPX_OK;
    if (!d_in || !d_out) return fail(DPX_ERR_ARG, "null device pointer");
    if (((uintptr_t)d_in | (uintptr_t)d_out) & 15u) return fail(DPX_ERR_ARG, "device pointers must be 16-byte aligned");
    // the device's lock for the whole enqueue: a resident block kernel (which would hold up this launch's queue) is asked to
    // leave, and no other thread's context starts one before the launches are in the queue (~40 ns when nothing contends)
    DPX_ENTER(plan->ctx);
    return run_plan(plan->host, plan->dev, d_in, in_fmt, d_out, out_fmt, plan->fma, plan->geom, hip_stream);
}

}  // extern "C"
