// dpx_context.cpp — context (one GPU), tuning knobs, plan images on the device, memory helpers
// (one of the translation units behind include/doppler_hip*.h: see dpx_internal.h)
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <new>

#include "dpx_internal.h"

namespace dpx_api {

namespace {
thread_local char g_err[512] = "";
}

DeviceState &device_state(int device)
{
    static DeviceState states[64];                 // hipGetDeviceCount of any node there is; index wraps for safety
    return states[(unsigned)device % 64u];
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int ensure_stage(dpx_ctx *ctx, size_t in_bytes, size_t out_bytes)
{
    if (in_bytes > ctx->stage_in_cap) {
        if (ctx->stage_in) DPX_HIP(hipFree(ctx->stage_in));
        ctx->stage_in = nullptr;
        ctx->stage_in_cap = 0;
        const size_t cap = in_bytes + in_bytes / 2 + 4096;
        DPX_HIP(hipMalloc(&ctx->stage_in, cap));
        ctx->stage_in_cap = cap;
    }
    if (out_bytes > ctx->stage_out_cap) {
        if (ctx->stage_out) DPX_HIP(hipFree(ctx->stage_out));
        ctx->stage_out = nullptr;
        ctx->stage_out_cap = 0;
        const size_t cap = out_bytes + out_bytes / 2 + 4096;
        DPX_HIP(hipMalloc(&ctx->stage_out, cap));
        ctx->stage_out_cap = cap;
    }
    return DPX_OK;
}

dpx::LaunchGeom geometry(const dpx_ctx *ctx)
{
    dpx::LaunchGeom g;
    g.block = ctx->block;
    g.vecs = ctx->vecs;
    g.autosel = ctx->geom_auto ? 1 : 0;
    g.legacy_cast = ctx->i16_cast == DPX_CAST_LEGACY_X86 ? 1 : 0;
    return g;
}

int choice_of(int variant)
{
    return variant == 4 ? dpx::kChooseTileOnly : variant == 5 ? dpx::kChooseWalk : variant == 6 ? dpx::kChooseRows
                                                                                                 : dpx::kChooseAuto;
}

dpx::PlanTuning tuning_of(const dpx_options *o)
{
    dpx::PlanTuning t;
    if (!o) return t;
    t.rows_mult = o->rows_mult;
    t.rows_maxl = o->rows_maxl;
    t.rows_r = o->rows_r;
    t.rows_compute = o->rows_compute;
    t.walk_waves = o->walk_waves;
    t.walk_tilemin = o->walk_tilemin;
    t.walk_span = o->walk_span;
    t.walk_flags = o->walk_flags;
    t.sub_lg = o->sub_lg;
    return t;
}

// the stretch list of a segment list (counter carried from segment to segment), periods scanned in parallel first
void append_segments(dpx::PlanResult &plan, const dpx_segment *segs, size_t n_segs, uint32_t samplerate, uint32_t &sn,
                     int variant, dpx::PeriodCache &cache)
{
    if (n_segs >= 16) {
        std::vector<float> ratios(n_segs);
        std::vector<uint64_t> counts(n_segs);
        for (size_t i = 0; i < n_segs; ++i) {
            ratios[i] = dpx::ratio_of(segs[i].shift_hz, samplerate);
            counts[i] = segs[i].n_samples;
        }
        cache.prefetch(ratios.data(), counts.data(), n_segs);
    }
    for (size_t i = 0; i < n_segs; ++i)
        dpx::plan_append(plan, dpx::ratio_of(segs[i].shift_hz, samplerate), segs[i].n_samples, sn, variant, &cache);
}

// Upload stretch + hint tables and fill the corrector tables (async on `st`).
// `plan` must have been finalize()d for geometry `g`.
int materialize(dpx_ctx *ctx, const dpx::PlanResult &plan, DevPlan &dev, bool fma, hipStream_t st)
{
    // device buffer: stretches | hints | leftover ranges | their hints | spans | span index   (one upload)
    //                | span descriptors, one per group of 8 workgroups (written on the device) | corrector tables
    const size_t seg_bytes = align256(plan.segs.size() * sizeof(dpx::DevSeg));
    const size_t hint_bytes = align256(plan.hint.size() * sizeof(uint32_t));
    const size_t left_bytes = align256(plan.left.size() * sizeof(dpx::LeftRange));
    const size_t lhint_bytes = align256(plan.left_hint.size() * sizeof(uint32_t));
    const size_t n_wdesc = plan.walk.empty() ? 0 : plan.walk_hint.size();
    const size_t span_bytes = align256(n_wdesc ? plan.walk.size() * sizeof(dpx::WalkSeg) : 0);
    const size_t index_bytes = align256(n_wdesc * sizeof(uint32_t));
    const size_t walk_bytes = align256(n_wdesc * sizeof(dpx::WalkSeg));
    const size_t lut_bytes = align256(plan.lut_entries * 8 + 64);
    const size_t image_bytes = seg_bytes + hint_bytes + left_bytes + lhint_bytes + span_bytes + index_bytes;
    const size_t need = image_bytes + walk_bytes + lut_bytes;
    if (need > dev.cap) {
        if (dev.buf) {
            DPX_HIP(hipStreamSynchronize(st));
            DPX_HIP(hipFree(dev.buf));
        }
        dev.buf = nullptr;
        dev.cap = 0;
        const size_t cap = need + need / 2;
        DPX_HIP(hipMalloc(&dev.buf, cap));
        dev.cap = cap;
    }
    char *base = static_cast<char *>(dev.buf);
    char *p = base;
    dev.segs = reinterpret_cast<dpx::DevSeg *>(p);           p += seg_bytes;
    dev.hint = reinterpret_cast<uint32_t *>(p);              p += hint_bytes;
    dev.left = reinterpret_cast<dpx::LeftRange *>(p);        p += left_bytes;
    dev.left_hint = reinterpret_cast<uint32_t *>(p);         p += lhint_bytes;
    char *d_spans = p;                                       p += span_bytes;
    char *d_index = p;                                       p += index_bytes;
    dev.walk = reinterpret_cast<dpx::WalkSeg *>(p);          p += walk_bytes;
    dev.lut = p;
    // one host image of all the small tables, one copy (every hipMemcpyAsync from pageable memory costs 5-8 us)
    dev.image.assign(image_bytes, 0);
    char *img = dev.image.data();
    auto put = [&](size_t off, const void *src, size_t bytes) { if (bytes) memcpy(img + off, src, bytes); };
    size_t off = 0;
    put(off, plan.segs.data(), plan.segs.size() * sizeof(dpx::DevSeg));              off += seg_bytes;
    put(off, plan.hint.data(), plan.hint.size() * sizeof(uint32_t));                 off += hint_bytes;
    put(off, plan.left.data(), plan.left.size() * sizeof(dpx::LeftRange));           off += left_bytes;
    put(off, plan.left_hint.data(), plan.left_hint.size() * sizeof(uint32_t));       off += lhint_bytes;
    if (n_wdesc) {
        put(off, plan.walk.data(), plan.walk.size() * sizeof(dpx::WalkSeg));         off += span_bytes;
        put(off, plan.walk_hint.data(), n_wdesc * sizeof(uint32_t));
    }
    DPX_HIP(hipMemcpyAsync(base, img, image_bytes, hipMemcpyHostToDevice, st));
    // the descriptor of every group of 8 workgroups, so that a workgroup finds its own with one scalar load: spans[index[group]]
    if (n_wdesc && dpx::launch_expand_walk(d_spans, d_index, dev.walk, (uint32_t)n_wdesc, st) != DPX_OK)
        return fail(DPX_ERR_HIP, "descriptor launch failed: %s", hipGetErrorString(hipGetLastError()));
    for (const dpx::TableBuild &t : plan.tables) {
        int rc = dpx::launch_build_lut(static_cast<char *>(dev.lut) + (size_t)t.off * 8, t.period, t.n_first,
                                       t.n_entries, t.ratio, fma, st);
        if (rc != DPX_OK) return fail(rc, "table build launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    return DPX_OK;
}

// every launch of a finalized plan, asynchronously on `st`
int run_plan(const dpx::PlanResult &plan, const DevPlan &dev, const void *d_in, int in_fmt, void *d_out,
             int out_fmt, bool fma, const dpx::LaunchGeom &g_in, void *st)
{
    // Tile-kernel geometry when the caller named none (all three are 1024-sample tiles: the plan fits any): 256 lanes x one
    // vector for f32 output and for tile tables; 128 lanes x two vectors for f32 -> i16; ONE wavefront x four vectors for
    // i16 -> i16, whose per-sample path is bound by vector instructions: 8 of the 45 per sample were the tile's set-up
    // (stretch lookup, phase of the tile, addresses), and sixteen samples per lane halve them.  Round 4, sincos per sample,
    // 3 Hz / 5001 Hz on one box (profiles/raw/r04_ab_persample4.log): i16 -> i16 64x4 73 / 65 % (128x2 66 / 61, 256x1 59 / 57);
    // f32 -> f32 256x1 78 / 78 (128x2 76 / 76); f32 -> i16 128x2 78.5 / 79.5 (256x1 79.5 / 75.5); i16 -> f32 256x1 80 / 76
    // (128x2 75 / 75).
    dpx::LaunchGeom g = g_in;
    const std::vector<dpx::Launch> &launches = dpx::launches_for(plan, in_fmt, out_fmt);
    if (g.autosel && g.tile() == 1024u) {
        // (tile tables were laid out for the tile ranges of the DEFAULT launch list: the whole-stream alternative of the
        // mixed pairs evaluates nearly everything per sample and takes the per-sample shapes)
        const bool wide = out_fmt == DPX_FMT_F32 || (plan.tile_tables && &launches == &plan.launches);
        g.block = wide ? 256 : in_fmt == DPX_FMT_I16 ? 64 : 128;
        g.vecs = wide ? 1 : in_fmt == DPX_FMT_I16 ? 4 : 2;
    }
    if (g.block == 64 && !(in_fmt == DPX_FMT_I16 && out_fmt == DPX_FMT_I16)) { g.block = 128; g.vecs = 2; }   // 64 x 4 exists for i16 -> i16 only
    for (const dpx::Launch &ln : launches) {
        int rc;
        if (ln.kind == 0)
            rc = dpx::launch_rows(d_in, in_fmt, d_out, out_fmt, dev.segs, dev.lut, ln.rows, fma, g.legacy_cast, st);
        else if (ln.kind == 2)
            rc = dpx::launch_span(d_in, in_fmt, d_out, out_fmt, dev.segs, dev.walk, dev.left, dev.left_hint, ln.walk, fma, g.legacy_cast, st);
        else
            rc = dpx::launch_tiles(d_in, in_fmt, d_out, out_fmt, dev.segs, (uint32_t)plan.segs.size(), dev.hint,
                                   dev.lut, ln.tiles, fma, g, st);
        if (rc != DPX_OK) return fail(rc, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    return DPX_OK;
}

void release(DevPlan &dev)
{
    if (dev.buf) (void)hipFree(dev.buf);
    dev = DevPlan();
}

}  // namespace dpx_api

using namespace dpx_api;

extern "C" {

int dpx_abi_version(void) { return DPX_ABI_VERSION; }

const char *dpx_last_error(void) { return g_err; }

int dpx_device_count(int *count)
{
    if (!count) return fail(DPX_ERR_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(DPX_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return DPX_OK;
}

int dpx_ctx_create(int device, dpx_ctx **out)
{
    if (!out) return fail(DPX_ERR_ARG, "ctx is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(DPX_ERR_NO_DEVICE, "no HIP device visible (%s); this library has no CPU path",
                    e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(DPX_ERR_NO_DEVICE, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    DPX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DPX_ERR_NO_DEVICE, "device %d is %s; kernels are built for gfx950 only", device,
                    prop.gcnArchName);
    DPX_HIP(hipSetDevice(device));
    dpx_ctx *ctx = new (std::nothrow) dpx_ctx;
    if (!ctx) return fail(DPX_ERR_ARG, "out of host memory");
    ctx->device = device;
    ctx->dev = &device_state(device);
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *e = getenv("DPX_RESIDENT")) ctx->resident_on = atoi(e) != 0;
    hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete ctx;
        return fail(DPX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se));
    }
    // Load the kernels' code object now (the HIP runtime does it at the first launch, ~8 ms): a first plan or a first
    // 8 KiB block should not pay for it.  A 16-byte copy inside a scratch allocation is the cheapest launch there is.
    void *warm = nullptr;
    if (hipMalloc(&warm, 1 << 20) == hipSuccess) {
        (void)dpx::launch_copy(warm, static_cast<char *>(warm) + 32, 16, ctx->stream);
        (void)dpx::launch_build_lut(warm, 4, 1, 4, 0.25f, ctx->fma, ctx->stream);
        // and the runtime's staging path for a copy from pageable memory (a plan's image: tens of KiB; 6.4 ms the first time)
        std::vector<char> image(64 << 10, 0);
        (void)hipMemcpyAsync(warm, image.data(), image.size(), hipMemcpyHostToDevice, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(warm);
    }
    *out = ctx;
    return DPX_OK;
}

void dpx_ctx_destroy(dpx_ctx *ctx)
{
    if (!ctx) return;
    std::lock_guard<std::recursive_mutex> lock(ctx->dev->mu);
    (void)hipSetDevice(ctx->device);
    // the resident block kernel leaves before its slots are freed (any context's: hipFree waits for the device); one that
    // does not answer may still be queued behind other work and poll them later — nothing of the context is freed under a
    // kernel that has not finished
    if (resident_stop_device(ctx) != DPX_OK || resident_stop(ctx) != DPX_OK) (void)hipDeviceSynchronize();
    // whatever happened above, the process-wide device state never points at a context that is about to be freed
    {
        dpx_ctx *me = ctx;
        ctx->dev->resident_owner.compare_exchange_strong(me, nullptr);
        ctx->resident_running.store(false, std::memory_order_release);
    }
    if (ctx->rstream) { (void)hipStreamSynchronize(ctx->rstream); (void)hipStreamDestroy(ctx->rstream); }
    if (ctx->rshared) (void)hipFree(ctx->rshared);
    if (ctx->stage_in) (void)hipFree(ctx->stage_in);
    if (ctx->stage_out) (void)hipFree(ctx->stage_out);
    if (ctx->scratch) {
        release(*ctx->scratch);
        delete ctx->scratch;
    }
    if (ctx->small_host) (void)hipHostFree(ctx->small_host);
    for (dpx_ctx::AsyncSlot &a : ctx->async_slots) {
        if (a.done) { (void)hipEventSynchronize(a.done); (void)hipEventDestroy(a.done); }
        if (a.host) (void)hipHostFree(a.host);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int dpx_set_tuning(dpx_ctx *ctx, int block, int vecs, int variant)
{
    if (!ctx) return fail(DPX_ERR_ARG, "ctx is null");
    if (variant < 0 || variant > 6) return fail(DPX_ERR_ARG, "variant out of range");
    if (block == -1 && vecs == -1) {               // back to the per-launch choice
        ctx->block = 128;
        ctx->vecs = 2;
        ctx->geom_auto = true;
        block = vecs = 0;
    }
    if (block != 0 && block != 64 && block != 128 && block != 256) return fail(DPX_ERR_ARG, "block must be 64, 128 or 256");
    if (vecs != 0 && vecs != 1 && vecs != 2 && vecs != 4) return fail(DPX_ERR_ARG, "vecs must be 1, 2 or 4");
    {   // the three 1024-sample tiles that are built: 256 lanes x 1 vector, 128 x 2, 64 x 4 (naming one half picks the other to match)
        int b = block ? block : (vecs ? 256 / vecs : ctx->block);
        int v = vecs ? vecs : (block ? 256 / block : ctx->vecs);
        if (b * v != 256) return fail(DPX_ERR_ARG, "tile geometry must be 256 x 1, 128 x 2 or 64 x 4");
        if (block || vecs) { ctx->block = b; ctx->vecs = v; ctx->geom_auto = false; }
    }
    ctx->choice = choice_of(variant);
    ctx->variant = variant >= 3 ? 0 : variant;
    return DPX_OK;
}

int dpx_set_options(dpx_ctx *ctx, const dpx_options *opt)
{
    if (!ctx) return fail(DPX_ERR_ARG, "ctx is null");
    if (opt) {
        if (opt->rows_r != 0 && opt->rows_r != 2 && opt->rows_r != 4) return fail(DPX_ERR_ARG, "rows_r must be 2 or 4");
        const uint32_t ww = opt->walk_waves;
        if (ww != 0 && !dpx::walk_waves_ok(ww, false)) return fail(DPX_ERR_ARG, "walk_waves must be 2, 4, 5 or 8");
        if (opt->walk_span == 1 || opt->walk_span > 4096) return fail(DPX_ERR_ARG, "walk_span must be 0 (the planner's cut) or 2..4096 rows");
        if (opt->sub_lg != 0 && opt->sub_lg < 12) return fail(DPX_ERR_ARG, "sub_lg must be 0 (the default), 12..47, or >= 48 (one launch whatever the length)");
    }
    ctx->tuning = tuning_of(opt);
    return DPX_OK;
}

int dpx_set_i16_cast(dpx_ctx *ctx, int mode)
{
    if (!ctx) return fail(DPX_ERR_ARG, "ctx is null");
    if (mode != DPX_CAST_SATURATE && mode != DPX_CAST_LEGACY_X86) return fail(DPX_ERR_ARG, "unknown i16 cast mode %d", mode);
    ctx->i16_cast = mode;
    return DPX_OK;
}

int dpx_set_libm_contraction(dpx_ctx *ctx, int fma)
{
    if (!ctx) return fail(DPX_ERR_ARG, "ctx is null");
    ctx->fma = fma != 0;
    return DPX_OK;
}

int dpx_debug_copy(dpx_ctx *ctx, const void *d_in, void *d_out, size_t n_bytes, void *hip_stream)
{
    if (!ctx || !d_in || !d_out || (n_bytes & 15u)) return fail(DPX_ERR_ARG, "bad argument");
    int rc = dpx::launch_copy(d_in, d_out, n_bytes, hip_stream);
    if (rc != DPX_OK) return fail(rc, "kernel launch failed");
    return DPX_OK;
}

/* -------------------------------------------------------------- memory helpers */

int dpx_malloc(dpx_ctx *ctx, size_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return fail(DPX_ERR_ARG, "bad argument");
    DPX_ENTER(ctx);
    DPX_HIP(hipMalloc(d_ptr, bytes ? bytes : 16));
    return DPX_OK;
}

int dpx_free(dpx_ctx *ctx, void *d_ptr)
{
    if (!ctx) return fail(DPX_ERR_ARG, "bad argument");
    DPX_ENTER(ctx);               // hipFree waits for the device: a resident block kernel leaves first
    if (d_ptr) DPX_HIP(hipFree(d_ptr));
    return DPX_OK;
}

int dpx_memcpy_h2d(dpx_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    if (!ctx) return fail(DPX_ERR_ARG, "bad argument");
    DPX_ENTER(ctx);
    if (bytes) DPX_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return DPX_OK;
}

int dpx_memcpy_d2h(dpx_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    if (!ctx) return fail(DPX_ERR_ARG, "bad argument");
    DPX_ENTER(ctx);
    if (bytes) DPX_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return DPX_OK;
}

int dpx_synchronize(dpx_ctx *ctx)
{
    if (!ctx) return fail(DPX_ERR_ARG, "bad argument");
    DPX_ENTER(ctx);
    DPX_HIP(hipDeviceSynchronize());
    return DPX_OK;
}

}  // extern "C"
