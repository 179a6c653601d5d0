// dpx_api.cpp — the C ABI of include/doppler_hip.h.
//
// Host side of the MI355X hot path: context (one GPU), staging for the
// host-pointer operator entry points, the plan objects, and the launches.
// There is deliberately no CPU implementation of any entry point here: if the
// GPU or the kernels are unavailable the calls fail with an error code.
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/doppler_hip.h"
#include "dpx_planner.h"
#include "dpx_types.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define DPX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(DPX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));    \
    } while (0)

inline size_t bytes_per_sample(int fmt) { return fmt == DPX_FMT_I16 ? 4 : 8; }
inline bool fmt_ok(int fmt) { return fmt == DPX_FMT_I16 || fmt == DPX_FMT_F32; }

}  // namespace

struct dpx_ctx {
    int device = -1;
    int n_cu = 0;
    bool fma = true;          // libm variant whose sincosf the kernels reproduce
    int blocks_per_cu = 8;
    int unroll = 4;
    int variant = 0;
    hipStream_t stream = nullptr;   // internal stream of the host-pointer entry points
    void *stage_in = nullptr;
    void *stage_out = nullptr;
    size_t stage_in_cap = 0, stage_out_cap = 0;
};

struct dpx_plan {
    dpx_ctx *ctx = nullptr;
    dpx::PlanResult host;
    dpx::DevSeg *d_segs = nullptr;
};

namespace {

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

dpx::LaunchGeom geometry(const dpx_ctx *ctx, const dpx::PlanResult &plan)
{
    dpx::LaunchGeom g;
    g.unroll = ctx->unroll;
    const uint64_t tile = (uint64_t)dpx::kBlock * dpx::kSamplesPerLane * g.unroll;
    const uint64_t n_tiles = (plan.n_samples + tile - 1) / tile;
    const uint64_t cap = (uint64_t)ctx->n_cu * ctx->blocks_per_cu;
    g.grid = (int)(n_tiles < 1 ? 1 : (n_tiles < cap ? n_tiles : cap));
    g.lds_bytes = plan.max_lut_len * (uint32_t)sizeof(float) * 2u;
    return g;
}

int upload_plan(dpx_plan *p)
{
    const size_t bytes = p->host.segs.size() * sizeof(dpx::DevSeg);
    if (bytes == 0) return DPX_OK;
    DPX_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_segs), bytes));
    DPX_HIP(hipMemcpy(p->d_segs, p->host.segs.data(), bytes, hipMemcpyHostToDevice));
    return DPX_OK;
}

// shared body of the host-pointer operators: stage in, one fused launch, stage out
int run_host(dpx_ctx *ctx, const void *in, size_t n, int in_fmt, void *out, int out_fmt,
             uint32_t *samplenum, float shift_hz, uint32_t samplerate)
{
    DPX_HIP(hipSetDevice(ctx->device));
    dpx::PlanResult plan;
    uint32_t sn = *samplenum;
    dpx::plan_append(plan, dpx::ratio_of(shift_hz, samplerate), n, sn, ctx->variant);
    if (n == 0) {
        *samplenum = sn;
        return DPX_OK;
    }
    const size_t in_bytes = n * bytes_per_sample(in_fmt), out_bytes = n * bytes_per_sample(out_fmt);
    const size_t seg_bytes = plan.segs.size() * sizeof(dpx::DevSeg);
    int rc = ensure_stage(ctx, in_bytes + 64 + seg_bytes, out_bytes);
    if (rc != DPX_OK) return rc;
    // stretch table rides behind the input in the same staging buffer (32-byte aligned)
    const size_t seg_off = (in_bytes + 63) & ~(size_t)63;
    dpx::DevSeg *d_segs = reinterpret_cast<dpx::DevSeg *>(static_cast<char *>(ctx->stage_in) + seg_off);
    DPX_HIP(hipMemcpyAsync(ctx->stage_in, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    DPX_HIP(hipMemcpyAsync(d_segs, plan.segs.data(), seg_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = dpx::launch_shift(ctx->stage_in, in_fmt, ctx->stage_out, out_fmt, d_segs,
                           (uint32_t)plan.segs.size(), plan.n_samples, ctx->fma,
                           geometry(ctx, plan), ctx->stream);
    if (rc != DPX_OK) return fail(rc, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    DPX_HIP(hipMemcpyAsync(out, ctx->stage_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    DPX_HIP(hipStreamSynchronize(ctx->stream));
    *samplenum = sn;
    return DPX_OK;
}

}  // namespace

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
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete ctx;
        return fail(DPX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se));
    }
    *out = ctx;
    return DPX_OK;
}

void dpx_ctx_destroy(dpx_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stage_in) (void)hipFree(ctx->stage_in);
    if (ctx->stage_out) (void)hipFree(ctx->stage_out);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int dpx_set_tuning(dpx_ctx *ctx, int blocks_per_cu, int unroll, int variant)
{
    if (!ctx) return fail(DPX_ERR_ARG, "ctx is null");
    if (blocks_per_cu < 0 || blocks_per_cu > 64) return fail(DPX_ERR_ARG, "blocks_per_cu out of range");
    if (unroll != 0 && unroll != 1 && unroll != 2 && unroll != 4 && unroll != 8)
        return fail(DPX_ERR_ARG, "unroll must be 1, 2, 4 or 8");
    if (variant < 0 || variant > 3) return fail(DPX_ERR_ARG, "variant out of range");
    if (blocks_per_cu) ctx->blocks_per_cu = blocks_per_cu;
    if (unroll) ctx->unroll = unroll;
    ctx->variant = variant == 3 ? 0 : variant;
    return DPX_OK;
}

int dpx_set_libm_contraction(dpx_ctx *ctx, int fma)
{
    if (!ctx) return fail(DPX_ERR_ARG, "ctx is null");
    ctx->fma = fma != 0;
    return DPX_OK;
}

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
    int rc = run_host(ctx, in, n, in_fmt, out, out_fmt, samplenum, shift_hz, samplerate);
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
    DPX_HIP(hipSetDevice(ctx->device));
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
    DPX_HIP(hipSetDevice(ctx->device));
    int rc = ensure_stage(ctx, n * 8, n * 4);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(ctx->stage_in, inbuf, n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = dpx::launch_pack_i16(ctx->stage_in, ctx->stage_out, n, ctx->stream);
    if (rc != DPX_OK) return fail(rc, "kernel launch failed");
    DPX_HIP(hipMemcpyAsync(out, ctx->stage_out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    DPX_HIP(hipStreamSynchronize(ctx->stream));
    return DPX_OK;
}

int dpx_ccexpf_imag(dpx_ctx *ctx, dpx_complex32 *z, size_t n)
{
    if (!ctx || (n && !z)) return fail(DPX_ERR_ARG, "bad argument");
    if (n == 0) return DPX_OK;
    DPX_HIP(hipSetDevice(ctx->device));
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

int dpx_plan_describe(const dpx_segment *segs, size_t n_segs, uint32_t samplerate,
                      uint32_t samplenum0, int variant, dpx_stretch *out, size_t cap,
                      size_t *n_out, uint32_t *final_samplenum)
{
    if ((n_segs && !segs) || !n_out || (cap && !out)) return fail(DPX_ERR_ARG, "bad argument");
    static_assert(sizeof(dpx_stretch) == sizeof(dpx::DevSeg), "dpx_stretch mirrors DevSeg");
    dpx::PlanResult plan;
    uint32_t sn = samplenum0;
    for (size_t i = 0; i < n_segs; ++i)
        dpx::plan_append(plan, dpx::ratio_of(segs[i].shift_hz, samplerate), segs[i].n_samples, sn, variant);
    *n_out = plan.segs.size();
    for (size_t i = 0; i < plan.segs.size() && i < cap; ++i) memcpy(&out[i], &plan.segs[i], sizeof(dpx_stretch));
    if (final_samplenum) *final_samplenum = sn;
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
    uint32_t sn = samplenum0;
    p->host.final_samplenum = sn;
    for (size_t i = 0; i < n_segs; ++i)
        dpx::plan_append(p->host, dpx::ratio_of(segs[i].shift_hz, samplerate), segs[i].n_samples, sn,
                         ctx->variant);
    hipError_t e = hipSetDevice(ctx->device);
    int rc = e == hipSuccess ? upload_plan(p) : fail(DPX_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
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

int dpx_plan_final_samplenum(const dpx_plan *plan, uint32_t *samplenum)
{
    if (!plan || !samplenum) return fail(DPX_ERR_ARG, "bad argument");
    *samplenum = plan->host.final_samplenum;
    return DPX_OK;
}

void dpx_plan_destroy(dpx_plan *plan)
{
    if (!plan) return;
    if (plan->d_segs) {
        (void)hipSetDevice(plan->ctx->device);
        (void)hipFree(plan->d_segs);
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
    int rc = dpx::launch_shift(d_in, in_fmt, d_out, out_fmt, plan->d_segs, (uint32_t)plan->host.segs.size(),
                               plan->host.n_samples, plan->ctx->fma, geometry(plan->ctx, plan->host),
                               hip_stream);
    if (rc != DPX_OK) return fail(rc, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return DPX_OK;
}

int dpx_debug_copy(dpx_ctx *ctx, const void *d_in, void *d_out, size_t n_bytes, void *hip_stream)
{
    if (!ctx || !d_in || !d_out || (n_bytes & 15u)) return fail(DPX_ERR_ARG, "bad argument");
    const uint64_t n_vec = n_bytes / 16, tile = (uint64_t)dpx::kBlock * 4;
    const uint64_t tiles = (n_vec + tile - 1) / tile, cap = (uint64_t)ctx->n_cu * ctx->blocks_per_cu;
    const int grid = (int)(tiles < 1 ? 1 : (tiles < cap ? tiles : cap));
    int rc = dpx::launch_copy(d_in, d_out, n_bytes, grid, hip_stream);
    if (rc != DPX_OK) return fail(rc, "kernel launch failed");
    return DPX_OK;
}

/* -------------------------------------------------------------- memory helpers */

int dpx_malloc(dpx_ctx *ctx, size_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return fail(DPX_ERR_ARG, "bad argument");
    DPX_HIP(hipSetDevice(ctx->device));
    DPX_HIP(hipMalloc(d_ptr, bytes ? bytes : 16));
    return DPX_OK;
}

int dpx_free(dpx_ctx *ctx, void *d_ptr)
{
    if (!ctx) return fail(DPX_ERR_ARG, "bad argument");
    if (d_ptr) DPX_HIP(hipFree(d_ptr));
    return DPX_OK;
}

int dpx_memcpy_h2d(dpx_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    if (!ctx) return fail(DPX_ERR_ARG, "bad argument");
    if (bytes) DPX_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return DPX_OK;
}

int dpx_memcpy_d2h(dpx_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    if (!ctx) return fail(DPX_ERR_ARG, "bad argument");
    if (bytes) DPX_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return DPX_OK;
}

int dpx_synchronize(dpx_ctx *ctx)
{
    if (!ctx) return fail(DPX_ERR_ARG, "bad argument");
    DPX_HIP(hipSetDevice(ctx->device));
    DPX_HIP(hipDeviceSynchronize());
    return DPX_OK;
}

}  // extern "C"
