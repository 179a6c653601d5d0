// dpx_api.cpp — the C ABI of include/doppler_hip.h.
//
// Host side of the MI355X hot path: context (one GPU), staging for the
// host-pointer operator entry points, the plan objects, and the launches.
// There is deliberately no CPU implementation of any entry point here: if the
// GPU or the kernels are unavailable the calls fail with an error code.
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/doppler_hip.h"
#include "../../include/doppler_hip_debug.h"
#include "../../include/doppler_hip_host.h"
#include "dpx_planner.h"
#include "dpx_types.h"
#include "host/orbit.h"
#include "host/schedule.h"

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
    int block = 128;          // tile kernel: lanes per workgroup (128 or 256)
    int vecs = 2;             // tile kernel: 4-sample groups per lane (1 or 2)
    bool geom_auto = true;    // until dpx_set_tuning names a geometry: chosen per launch (run_plan)
    int variant = 0;
    int choice = dpx::kChooseAuto;   // which kernels finalize() may use (dpx_set_tuning)
    int i16_cast = DPX_CAST_SATURATE;   // meaning of `as i16` (dpx_set_i16_cast); the legacy one confines plans to the tile kernel
    dpx::PlanTuning tuning;          // kernel-shape knobs (dpx_set_options)
    dpx::PeriodCache periods;        // period per ratio seen so far (one producer thread plans at a time)
    hipStream_t stream = nullptr;   // internal stream of the host-pointer entry points
    void *stage_in = nullptr;
    void *stage_out = nullptr;
    size_t stage_in_cap = 0, stage_out_cap = 0;
    struct DevPlan *scratch = nullptr;   // device side of the host-pointer operators' plans
    // small calls (the reference's own 8 KiB block): one pinned, device-mapped host buffer holds input, output and
    // the plan image; the kernel reads and writes it over PCIe directly, so a call is memcpy + one launch + one wait
    char *small_host = nullptr;
    char *small_dev = nullptr;
    // dpx_shift_block_async / dpx_wait: a ring of such buffers, one per block in flight
    static constexpr int kAsyncSlots = 4;
    struct AsyncSlot {
        char *host = nullptr, *dev = nullptr;
        hipEvent_t done = nullptr;
        uint32_t seq = 0;            // ticket of the block the slot holds (0: free)
        size_t out_bytes = 0, n_samples = 0;
        bool resident = false;       // the block was handed to the resident kernel (completion word), not launched (event)
    } async_slots[kAsyncSlots];
    uint32_t async_next_seq = 1;
    // the resident block kernel (dpx_types.h, BlockCtl): one workgroup per slot, launched once, polling the slots' doorbells
    bool resident_on = true;         // DPX_RESIDENT=0 or dpx_set_resident(ctx, 0): every block is a launch, as in round 3
    bool resident_running = false;   // host's view: a kernel has been launched and not yet seen parked
    int resident_in = -1, resident_out = -1;
    bool resident_fma = true;
    hipStream_t rstream = nullptr;
    dpx::ResidentShared *rshared = nullptr;
    uint64_t resident_launches = 0, resident_blocks = 0;
};
static_assert(dpx_ctx::kAsyncSlots == dpx::kResidentSlots, "one resident workgroup per staging slot");

// device image of a plan: stretch table | hint table | corrector-table pool, one allocation
struct DevPlan {
    void *buf = nullptr;
    size_t cap = 0;
    dpx::DevSeg *segs = nullptr;
    uint32_t *hint = nullptr;
    dpx::WalkSeg *walk = nullptr;   // one descriptor per 2^kWalkHintShift workgroups of the walk launch
    dpx::LeftRange *left = nullptr;
    uint32_t *left_hint = nullptr;
    void *lut = nullptr;
    std::vector<char> image;       // host copy of everything before the tables, source of the one upload
};

struct dpx_plan {
    dpx_ctx *ctx = nullptr;
    dpx::PlanResult host;
    dpx::LaunchGeom geom;
    bool fma = true;
    DevPlan dev;
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

int plan_choice(const dpx_ctx *ctx) { return ctx->choice; }

dpx::LaunchGeom geometry(const dpx_ctx *ctx)
{
    dpx::LaunchGeom g;
    g.block = ctx->block;
    g.vecs = ctx->vecs;
    g.autosel = ctx->geom_auto ? 1 : 0;
    g.legacy_cast = ctx->i16_cast == DPX_CAST_LEGACY_X86 ? 1 : 0;
    return g;
}

// dpx_set_tuning variants 4..6 restrict the kernels a plan may use (measurement A/B)
inline int choice_of(int variant)
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

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Upload stretch + hint tables and fill the corrector tables (async on `st`).
// `plan` must have been finalize()d for geometry `g`.
int materialize(dpx_ctx *ctx, const dpx::PlanResult &plan, DevPlan &dev, bool fma, hipStream_t st)
{
    const size_t seg_bytes = align256(plan.segs.size() * sizeof(dpx::DevSeg));
    const size_t hint_bytes = align256(plan.hint.size() * sizeof(uint32_t));
    const size_t n_wdesc = plan.walk.empty() ? 0 : plan.walk_hint.size();
    const size_t walk_bytes = align256(n_wdesc * sizeof(dpx::WalkSeg));
    const size_t left_bytes = align256(plan.left.size() * sizeof(dpx::LeftRange));
    const size_t lhint_bytes = align256(plan.left_hint.size() * sizeof(uint32_t));
    const size_t lut_bytes = align256(plan.lut_entries * 8 + 64);
    const size_t need = seg_bytes + hint_bytes + walk_bytes + left_bytes + lhint_bytes + lut_bytes;
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
    dev.segs = reinterpret_cast<dpx::DevSeg *>(base);
    dev.hint = reinterpret_cast<uint32_t *>(base + seg_bytes);
    char *p = base + seg_bytes + hint_bytes;
    dev.walk = reinterpret_cast<dpx::WalkSeg *>(p);          p += walk_bytes;
    dev.left = reinterpret_cast<dpx::LeftRange *>(p);        p += left_bytes;
    dev.left_hint = reinterpret_cast<uint32_t *>(p);         p += lhint_bytes;
    dev.lut = p;
    // one host image of all the small tables, one copy (every hipMemcpyAsync from pageable memory costs 5-8 us)
    const size_t image_bytes = seg_bytes + hint_bytes + walk_bytes + left_bytes + lhint_bytes;
    dev.image.assign(image_bytes, 0);
    char *img = dev.image.data();
    auto put = [&](size_t off, const void *src, size_t bytes) { if (bytes) memcpy(img + off, src, bytes); };
    size_t off = 0;
    put(off, plan.segs.data(), plan.segs.size() * sizeof(dpx::DevSeg));              off += seg_bytes;
    put(off, plan.hint.data(), plan.hint.size() * sizeof(uint32_t));                 off += hint_bytes;
    // the chunk descriptor of every group of 8 workgroups, so that a workgroup finds its own with one scalar load
    for (size_t h = 0; h < n_wdesc; ++h) memcpy(img + off + h * sizeof(dpx::WalkSeg), &plan.walk[plan.walk_hint[h]], sizeof(dpx::WalkSeg));
    off += walk_bytes;
    put(off, plan.left.data(), plan.left.size() * sizeof(dpx::LeftRange));           off += left_bytes;
    put(off, plan.left_hint.data(), plan.left_hint.size() * sizeof(uint32_t));
    DPX_HIP(hipMemcpyAsync(base, img, image_bytes, hipMemcpyHostToDevice, st));
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
    if (g.autosel && g.tile() == 1024u) {
        const bool wide = out_fmt == DPX_FMT_F32 || plan.tile_tables;
        g.block = wide ? 256 : in_fmt == DPX_FMT_I16 ? 64 : 128;
        g.vecs = wide ? 1 : in_fmt == DPX_FMT_I16 ? 4 : 2;
    }
    if (g.block == 64 && !(in_fmt == DPX_FMT_I16 && out_fmt == DPX_FMT_I16)) { g.block = 128; g.vecs = 2; }   // 64 x 4 exists for i16 -> i16 only
    for (const dpx::Launch &ln : dpx::launches_for(plan, in_fmt, out_fmt)) {
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

constexpr size_t kSmallCallBytes = 64 << 10;      // per side; larger calls go through device staging buffers
constexpr size_t kSmallPlanBytes = 16 << 10;
constexpr size_t kSmallInOff = 0, kSmallOutOff = kSmallCallBytes, kSmallPlanOff = 2 * kSmallCallBytes;
constexpr size_t kSmallCtlOff = 2 * kSmallCallBytes + kSmallPlanBytes;      // dpx::BlockCtl of an asynchronous slot
constexpr size_t kSlotBytes = kSmallCtlOff + 256;
constexpr uint64_t kResidentIdleTicks = 200000;       // 2 ms of the 100 MHz wall clock without a block: the kernel leaves
constexpr double kResidentTimeoutS = 5.0;             // a completion word that does not come: error, resident mode off

inline dpx::BlockCtl *slot_ctl(dpx_ctx::AsyncSlot &a) { return reinterpret_cast<dpx::BlockCtl *>(a.host + kSmallCtlOff); }
inline uint32_t load_acq(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void store_rel(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline double mono_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void cpu_relax()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

int alloc_slot(dpx_ctx::AsyncSlot &a)
{
    if (a.host) return DPX_OK;
    void *h = nullptr, *d = nullptr;
    DPX_HIP(hipHostMalloc(&h, kSlotBytes, hipHostMallocMapped));
    hipError_t e = hipHostGetDevicePointer(&d, h, 0);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&a.done, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipHostFree(h);
        return fail(DPX_ERR_HIP, "asynchronous block slot: %s", hipGetErrorString(e));
    }
    memset(static_cast<char *>(h) + kSmallCtlOff, 0, 256);
    a.host = static_cast<char *>(h);
    a.dev = static_cast<char *>(d);
    slot_ctl(a)->state = dpx::kResidentParked;
    return DPX_OK;
}

bool resident_all_parked(dpx_ctx *ctx)
{
    for (auto &a : ctx->async_slots)
        if (a.host && load_acq(&slot_ctl(a)->state) != dpx::kResidentParked) return false;
    return true;
}

// Ask the resident kernel to leave and wait until it has: before any launch of this context's own (a resident kernel holds
// its hardware queue; another stream's launch that shares the queue would wait for it), before the context goes away, and
// when the format pair changes.  Blocks already rung are finished first.  Costs one load when no kernel is running.
int resident_stop(dpx_ctx *ctx)
{
    if (!ctx->resident_running) return DPX_OK;
    const double t0 = mono_s();
    for (auto &a : ctx->async_slots) {             // blocks in flight: their completion words first
        if (!a.host || !a.resident || a.seq == 0) continue;
        dpx::BlockCtl *c = slot_ctl(a);
        while (load_acq(&c->done) != a.seq && load_acq(&c->state) != dpx::kResidentParked && mono_s() - t0 < kResidentTimeoutS) cpu_relax();
    }
    for (auto &a : ctx->async_slots) {
        if (!a.host) continue;
        dpx::BlockCtl *c = slot_ctl(a);
        if (a.resident && a.seq != 0 && load_acq(&c->done) != a.seq) continue;      // rung but unserved (the kernel left first): keep the ticket in the doorbell
        store_rel(&c->doorbell, dpx::kDoorExit);
    }
    // (a slot whose doorbell still holds an unserved ticket cannot be told to leave through it: the idle clock takes that
    // workgroup out within kResidentIdleTicks once the others have gone — it cannot happen unless the kernel left early)
    while (!resident_all_parked(ctx)) {
        if (mono_s() - t0 > kResidentTimeoutS) {
            ctx->resident_on = false;
            return fail(DPX_ERR_HIP, "the resident block kernel does not leave");
        }
        cpu_relax();
    }
    DPX_HIP(hipStreamSynchronize(ctx->rstream));
    ctx->resident_running = false;
    return DPX_OK;
}

// A resident kernel for (in_fmt, out_fmt) is polling every slot's doorbell when this returns.
int resident_ensure(dpx_ctx *ctx, int in_fmt, int out_fmt)
{
    if (ctx->resident_running) {
        const bool same = ctx->resident_in == in_fmt && ctx->resident_out == out_fmt && ctx->resident_fma == ctx->fma;
        bool any_parked = false;
        for (auto &a : ctx->async_slots) any_parked = any_parked || load_acq(&slot_ctl(a)->state) == dpx::kResidentParked;
        if (same && !any_parked) return DPX_OK;
        if (!same) {
            const int rc = resident_stop(ctx);
            if (rc != DPX_OK) return rc;
        } else {
            // the kernel is leaving (idle clock): its workgroups go within microseconds of each other
            const double t0 = mono_s();
            while (!resident_all_parked(ctx)) {
                if (mono_s() - t0 > kResidentTimeoutS) {
                    ctx->resident_on = false;
                    return fail(DPX_ERR_HIP, "the resident block kernel does not leave");
                }
                cpu_relax();
            }
            DPX_HIP(hipStreamSynchronize(ctx->rstream));
            ctx->resident_running = false;
        }
    }
    for (auto &a : ctx->async_slots) {
        const int rc = alloc_slot(a);
        if (rc != DPX_OK) return rc;
    }
    if (!ctx->rstream) DPX_HIP(hipStreamCreateWithFlags(&ctx->rstream, hipStreamNonBlocking));
    if (!ctx->rshared) DPX_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->rshared), sizeof(dpx::ResidentShared)));
    DPX_HIP(hipMemsetAsync(ctx->rshared, 0, sizeof(dpx::ResidentShared), ctx->rstream));
    dpx::ResidentArgs ra;
    for (int k = 0; k < dpx_ctx::kAsyncSlots; ++k) {
        dpx_ctx::AsyncSlot &a = ctx->async_slots[k];
        dpx::BlockCtl *c = slot_ctl(a);
        if (c->doorbell == dpx::kDoorExit) c->doorbell = c->done;       // a stop request of the past is not one for this launch
        store_rel(&c->state, dpx::kResidentRunning);
        ra.ctl[k] = reinterpret_cast<dpx::BlockCtl *>(a.dev + kSmallCtlOff);
        ra.in[k] = reinterpret_cast<const uint8_t *>(a.dev + kSmallInOff);
        ra.out[k] = reinterpret_cast<uint8_t *>(a.dev + kSmallOutOff);
        ra.segs[k] = reinterpret_cast<const dpx::DevSeg *>(a.dev + kSmallPlanOff);
    }
    ra.shared = ctx->rshared;
    ra.idle_ticks = kResidentIdleTicks;
    const int rc = dpx::launch_resident_block(ra, in_fmt, out_fmt, ctx->fma, ctx->rstream);
    if (rc != DPX_OK) {
        for (auto &a : ctx->async_slots) slot_ctl(a)->state = dpx::kResidentParked;
        return fail(rc, "resident block kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    ctx->resident_running = true;
    ctx->resident_in = in_fmt;
    ctx->resident_out = out_fmt;
    ctx->resident_fma = ctx->fma;
    ++ctx->resident_launches;
    return DPX_OK;
}

// every entry point that launches or synchronises on this context's behalf passes through here first
#define DPX_ENTER(ctx)                                              \
    do {                                                            \
        DPX_HIP(hipSetDevice((ctx)->device));                       \
        if ((ctx)->resident_running) {                              \
            const int rc_enter_ = resident_stop(ctx);               \
            if (rc_enter_ != DPX_OK) return rc_enter_;              \
        }                                                           \
    } while (0)

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
    dpx::finalize(plan, g.tile(), plan_choice(ctx), ctx->tuning);
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
    dpx::finalize(plan, g.tile(), plan_choice(ctx), ctx->tuning);
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
    (void)hipSetDevice(ctx->device);
    (void)resident_stop(ctx);                   // the resident block kernel leaves before its slots are freed
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
        if (opt->rows_r != 0 && opt->rows_r != 2 && opt->rows_r != 4 && opt->rows_r != 8) return fail(DPX_ERR_ARG, "rows_r must be 2, 4 or 8");
        const uint32_t ww = opt->walk_waves;
        if (ww != 0 && !dpx::walk_waves_ok(ww, false)) return fail(DPX_ERR_ARG, "walk_waves must be 2, 4, 5 or 8");
        if (opt->walk_span == 1 || opt->walk_span > 4096) return fail(DPX_ERR_ARG, "walk_span must be 0 (the planner's cut) or 2..4096 rows");
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

// ---- one block in flight while the caller reads the next (main.rs:113-118 with its read overlapped)
int dpx_shift_block_async(dpx_ctx *ctx, const void *in, size_t in_bytes, int in_fmt, int out_fmt, uint32_t *samplenum,
                          float shift_hz, uint32_t samplerate, dpx_ticket *ticket)
{
    if (!ctx || !samplenum || !ticket || (!in && in_bytes) || !fmt_ok(in_fmt) || !fmt_ok(out_fmt))
        return fail(DPX_ERR_ARG, "bad argument");
    if (in_bytes % bytes_per_sample(in_fmt) != 0)
        return fail(DPX_ERR_BLOCK_LEN, "%zu bytes is not a whole number of %s samples", in_bytes,
                    in_fmt == DPX_FMT_I16 ? "i16" : "f32");
    const size_t n = in_bytes / bytes_per_sample(in_fmt);
    if (n * 8 > kSmallCallBytes) return fail(DPX_ERR_CAPACITY, "an asynchronous block holds at most %zu samples", kSmallCallBytes / 8);
    DPX_HIP(hipSetDevice(ctx->device));
    const uint32_t seq = ctx->async_next_seq;
    dpx_ctx::AsyncSlot &a = ctx->async_slots[seq % dpx_ctx::kAsyncSlots];
    if (a.seq != 0) return fail(DPX_ERR_PLAN, "%d blocks are in flight: dpx_wait for ticket %u first", dpx_ctx::kAsyncSlots, a.seq);
    int rc = alloc_slot(a);
    if (rc != DPX_OK) return rc;
    dpx::PlanResult plan;
    uint32_t sn = *samplenum;
    dpx::plan_append(plan, dpx::ratio_of(shift_hz, samplerate), n, sn, 1 /* sincos per sample */, &ctx->periods);
    a.n_samples = n;
    a.out_bytes = n * bytes_per_sample(out_fmt);
    a.resident = false;
    auto issue = [&](uint32_t sn_after) {
        a.seq = seq;
        uint32_t next = seq + 1;
        if (next == 0 || next == dpx::kDoorExit) next = 1;           // 0 marks a free slot, kDoorExit asks the resident kernel to leave
        ctx->async_next_seq = next;
        *samplenum = sn_after;            // the counter after the block is known as soon as the block is planned
        *ticket = seq;
        return DPX_OK;
    };
    if (n == 0) {
        if (ctx->resident_running) { rc = resident_stop(ctx); if (rc != DPX_OK) return rc; }
        DPX_HIP(hipEventRecord(a.done, ctx->stream));
        return issue(sn);
    }
    bool tabulated = false;
    for (const dpx::DevSeg &sg : plan.segs) tabulated = tabulated || sg.lut_len != 0;
    if (tabulated) {
        // The block's plan wants a corrector table (periods below 4: shift 0, samplerate / 2 ...; the reference resets the
        // counter on every sample there): this block takes the synchronous path into the slot's output buffer — same bytes,
        // same ticket protocol, no overlap for this one block.
        uint32_t sn_sync = *samplenum;
        rc = run_host(ctx, in, n, in_fmt, a.host + kSmallOutOff, out_fmt, &sn_sync, shift_hz, samplerate);
        if (rc != DPX_OK) return rc;
        DPX_HIP(hipEventRecord(a.done, ctx->stream));
        return issue(sn_sync);
    }
    if (ctx->resident_on && plan.segs.size() <= dpx::kResidentMaxSegs) {
        // ---- the resident kernel: payload and stretch list into the slot, then the doorbell
        dpx::BlockCtl *c = slot_ctl(a);
        memcpy(a.host + kSmallInOff, in, in_bytes);
        memcpy(a.host + kSmallPlanOff, plan.segs.data(), plan.segs.size() * sizeof(dpx::DevSeg));
        c->n_samples = (uint32_t)n;
        c->n_segs = (uint32_t)plan.segs.size();
        c->legacy = (ctx->i16_cast == DPX_CAST_LEGACY_X86 && out_fmt == DPX_FMT_I16) ? 1u : 0u;
        rc = resident_ensure(ctx, in_fmt, out_fmt);
        if (rc == DPX_OK) {
            store_rel(&c->doorbell, seq);
            a.resident = true;
            ++ctx->resident_blocks;
            return issue(sn);
        }
        if (ctx->resident_on) return rc;              // (a kernel that does not answer turns the mode off: the launch path below)
    }
    // ---- one launch per block (round 3's path): periods the resident kernel's slot cannot hold, or resident mode off
    if (ctx->resident_running) { rc = resident_stop(ctx); if (rc != DPX_OK) return rc; }
    {
        const dpx::LaunchGeom g = geometry(ctx);
        dpx::finalize(plan, g.tile(), dpx::kChooseTileOnly);
        if (plan.error) return fail(DPX_ERR_PLAN, "%s", plan.error);
        const size_t seg_bytes = align256(plan.segs.size() * sizeof(dpx::DevSeg));
        const size_t hint_bytes = plan.hint.size() * sizeof(uint32_t);
        if (plan.lut_entries != 0 || seg_bytes + hint_bytes > kSmallPlanBytes) {
            uint32_t sn_sync = *samplenum;
            rc = run_host(ctx, in, n, in_fmt, a.host + kSmallOutOff, out_fmt, &sn_sync, shift_hz, samplerate);
            if (rc != DPX_OK) return rc;
            DPX_HIP(hipEventRecord(a.done, ctx->stream));
            return issue(sn_sync);
        }
        memcpy(a.host + kSmallInOff, in, in_bytes);
        memcpy(a.host + kSmallPlanOff, plan.segs.data(), plan.segs.size() * sizeof(dpx::DevSeg));
        memcpy(a.host + kSmallPlanOff + seg_bytes, plan.hint.data(), hint_bytes);
        DevPlan dev;
        dev.segs = reinterpret_cast<dpx::DevSeg *>(a.dev + kSmallPlanOff);
        dev.hint = reinterpret_cast<uint32_t *>(a.dev + kSmallPlanOff + seg_bytes);
        dev.lut = a.dev + kSmallPlanOff;      // never read: no tabulated stretch in this plan
        rc = run_plan(plan, dev, a.dev + kSmallInOff, in_fmt, a.dev + kSmallOutOff, out_fmt, ctx->fma, g, ctx->stream);
        if (rc != DPX_OK) return rc;
    }
    DPX_HIP(hipEventRecord(a.done, ctx->stream));
    return issue(sn);
}

int dpx_wait(dpx_ctx *ctx, dpx_ticket ticket, void *out, size_t out_cap, size_t *n_samples_out)
{
    if (!ctx || ticket == 0) return fail(DPX_ERR_ARG, "bad argument");
    dpx_ctx::AsyncSlot &a = ctx->async_slots[ticket % dpx_ctx::kAsyncSlots];
    if (a.seq != ticket) return fail(DPX_ERR_ARG, "ticket %u is not in flight", ticket);
    if (a.out_bytes > out_cap || (!out && a.out_bytes))
        return fail(DPX_ERR_CAPACITY, "output needs %zu bytes, capacity %zu", a.out_bytes, out_cap);
    if (a.resident) {
        // the block's completion word in host memory; a kernel that left meanwhile (idle clock) is launched again and
        // finds the doorbell rung
        dpx::BlockCtl *c = slot_ctl(a);
        const double t0 = mono_s();
        uint32_t spins = 0;
        while (load_acq(&c->done) != ticket) {
            if ((++spins & 63u) == 0) {
                if (load_acq(&c->state) == dpx::kResidentParked && load_acq(&c->done) != ticket) {
                    DPX_HIP(hipSetDevice(ctx->device));
                    const int rc = resident_ensure(ctx, ctx->resident_in, ctx->resident_out);
                    if (rc != DPX_OK) { a.seq = 0; return rc; }
                }
                if (mono_s() - t0 > kResidentTimeoutS) {
                    a.seq = 0;
                    ctx->resident_on = false;
                    return fail(DPX_ERR_HIP, "the resident block kernel did not finish ticket %u", ticket);
                }
            }
            cpu_relax();
        }
    } else {
        const hipError_t e = hipEventSynchronize(a.done);
        if (e != hipSuccess) {
            a.seq = 0;                  // the slot is free again whatever happened to its block
            return fail(DPX_ERR_HIP, "waiting for ticket %u: %s", ticket, hipGetErrorString(e));
        }
    }
    if (a.out_bytes) memcpy(out, a.host + kSmallOutOff, a.out_bytes);
    if (n_samples_out) *n_samples_out = a.n_samples;
    a.seq = 0;
    return DPX_OK;
}

int dpx_set_resident(dpx_ctx *ctx, int on)
{
    if (!ctx) return fail(DPX_ERR_ARG, "ctx is null");
    if (!on && ctx->resident_running) {
        const int rc = resident_stop(ctx);
        if (rc != DPX_OK) return rc;
    }
    ctx->resident_on = on != 0;
    return DPX_OK;
}

int dpx_resident_stats(const dpx_ctx *ctx, uint64_t *launches, uint64_t *blocks)
{
    if (!ctx) return fail(DPX_ERR_ARG, "ctx is null");
    if (launches) *launches = ctx->resident_launches;
    if (blocks) *blocks = ctx->resident_blocks;
    return DPX_OK;
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
    append_segments(p->host, segs, n_segs, samplerate, sn, ctx->variant, ctx->periods);
    dpx::finalize(p->host, p->geom.tile(), plan_choice(ctx), ctx->tuning);
    hipError_t e = hipSetDevice(ctx->device);
    int rc = e == hipSuccess ? DPX_OK : fail(DPX_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    if (rc == DPX_OK) rc = resident_stop(ctx);
    if (rc == DPX_OK && p->host.error) rc = fail(DPX_ERR_PLAN, "%s", p->host.error);
    if (rc == DPX_OK && p->host.n_samples) {
        rc = materialize(ctx, p->host, p->dev, p->fma, ctx->stream);
        if (rc == DPX_OK) {
            e = hipStreamSynchronize(ctx->stream);   // tables are complete before any user stream runs
            if (e != hipSuccess) rc = fail(DPX_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
        }
    }
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
    if (plan->dev.buf) {
        (void)hipSetDevice(plan->ctx->device);
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
    if (plan->ctx->resident_running) {           // (one load otherwise) a resident block kernel would hold up this launch's queue
        const int rc = resident_stop(plan->ctx);
        if (rc != DPX_OK) return rc;
    }
    return run_plan(plan->host, plan->dev, d_in, in_fmt, d_out, out_fmt, plan->fma, plan->geom, hip_stream);
}

/* ------------------------------------------------------------------ streaming */

// what a slab's resident plan was made from (dpx_stream_submit reuses plan and device image on a match)
struct SlabKey {
    uint64_t segs_hash = 0;
    uint32_t samplerate = 0, sn_start = 0;
    int variant = 0, choice = 0, fma = 0, cast = 0, block = 0, vecs = 0, autosel = 0;
    dpx::PlanTuning tuning;
    bool operator==(const SlabKey &o) const
    {
        return segs_hash == o.segs_hash && samplerate == o.samplerate && sn_start == o.sn_start && variant == o.variant &&
               choice == o.choice && fma == o.fma && cast == o.cast && block == o.block && vecs == o.vecs && autosel == o.autosel &&
               tuning == o.tuning;
    }
};

static SlabKey slab_key(const dpx_ctx *ctx, const dpx::LaunchGeom &g, const dpx_segment *segs, size_t n_segs, uint32_t samplerate, uint32_t sn)
{
    SlabKey k;
    uint64_t h = 1469598103934665603ull;                       // FNV-1a over the segments' fields (the list itself is compared as well)
    for (size_t i = 0; i < n_segs; ++i) {
        uint32_t bits;
        memcpy(&bits, &segs[i].shift_hz, sizeof bits);
        h = (h ^ segs[i].n_samples) * 1099511628211ull;
        h = (h ^ bits) * 1099511628211ull;
    }
    k.segs_hash = h;
    k.samplerate = samplerate;
    k.sn_start = sn;
    k.variant = ctx->variant;
    k.choice = ctx->choice;
    k.fma = ctx->fma;
    k.cast = ctx->i16_cast;
    k.block = g.block;
    k.vecs = g.vecs;
    k.autosel = g.autosel;
    k.tuning = ctx->tuning;
    return k;
}

struct dpx_stream_slab {
    SlabKey key;
    bool have_key = false;
    std::vector<dpx_segment> key_segs;
    uint32_t key_sn_after = 0;
    dpx_ctx *ctx = nullptr;      // the GPU this slab is processed on (slab k of the ring belongs to context k mod n)
    char *h_in = nullptr, *h_out = nullptr;
    void *d_in = nullptr, *d_out = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    int numa_node = -1;          // where the pinned buffers were placed (-1: the caller's default policy)
    // several GPUs: the device work of a slab is enqueued by its GPU's own thread (dpx_stream::Worker)
    std::atomic<int> enq{0};     // 0: nothing pending; 1: handed to the worker; 2: enqueued (enq_rc says how it went)
    int enq_rc = 0;
    std::string enq_err;
    size_t job_in_bytes = 0;
    bool job_reuse = false;
    dpx::LaunchGeom job_geom = {128, 2};
    bool job_fma = true;
    dpx::PlanResult plan;
    DevPlan dev;
    size_t out_bytes = 0;
    std::atomic<int> state{0};   // 0 free, 1 acquired (being filled), 2 in flight, 3 handed out by next()
    dpx_stream_slab() = default;
    dpx_stream_slab(const dpx_stream_slab &) {}   // slabs are only ever default-constructed (vector::resize)
};

struct dpx_stream {
    dpx_ctx *ctx = nullptr;      // first context: holds the period cache and the tuning all slabs are planned with
    std::vector<dpx_ctx *> ctxs;
    int in_fmt = 0, out_fmt = 0;
    uint32_t samplerate = 0, samplenum = 0;
    size_t slab_bytes = 0, slab_out = 0;
    std::vector<dpx_stream_slab> slabs;
    size_t acq = 0;     // next slab to acquire            (producer side: acquire, then submit in the same order)
    size_t head = 0;    // oldest acquired slab, the next to submit
    size_t tail = 0;    // oldest submitted slab not yet handed out   (consumer side: next)
    size_t rel = 0;     // oldest handed-out slab                     (release, in the same order)
    std::atomic<int> in_flight{0};
    dpx_stream_stats stats = {};   // host cost of dpx_stream_submit, by part (dpx_stream_get_stats)
    // Several GPUs: one enqueue thread per context.  dpx_stream_submit plans on the caller's thread (the counter is carried
    // from slab to slab: sequential by nature, 0.1-1 us) and hands the device work — plan image, H2D, launch, D2H, event:
    // 7-12 us of HIP calls — to the thread of the slab's GPU, so that eight GPUs are fed by eight threads, each running on
    // the NUMA node of its GPU, and a slow call into one GPU's runtime does not hold up the others.
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<size_t> jobs;
        bool stop = false;
    };
    std::vector<std::unique_ptr<Worker>> workers;
    std::mutex enq_mu;             // guards stats' upload / enqueue parts and wakes dpx_stream_next
    std::condition_variable enq_cv;
};

namespace {
void slab_worker(dpx_stream *s, dpx_stream::Worker *w, int numa_node);
}

namespace {

// NUMA node of a GPU's PCIe root (sysfs, through the device's PCI bus id), or -1.  An 8-GPU MI355X node has two sockets,
// four GPUs under each: a slab ring whose pinned buffers all come from the creating thread's node sends half of the
// D2H traffic (8 x 25-28 GB/s at the kernel's rate) across the socket link.
int gpu_numa_node(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) return -1;
    for (char *c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[128];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

// Pinned allocations of the calling thread prefer `node` until the policy is reset (node < 0: the default policy).
// set_mempolicy(2) by number: no libnuma in the image; failure (no NUMA, seccomp) is silent — the default policy stays.
void prefer_numa_node(int node)
{
#ifdef SYS_set_mempolicy
    constexpr int kMpolDefault = 0, kMpolPreferred = 1;
    if (node < 0 || node >= 1024) {
        (void)syscall(SYS_set_mempolicy, kMpolDefault, nullptr, 0);
        return;
    }
    unsigned long mask[1024 / (8 * sizeof(unsigned long))] = {0};
    mask[(size_t)node / (8 * sizeof(unsigned long))] |= 1ul << ((size_t)node % (8 * sizeof(unsigned long)));
    (void)syscall(SYS_set_mempolicy, kMpolPreferred, mask, 1024 + 1);
#else
    (void)node;
#endif
}

}  // namespace

int dpx_stream_create_multi(dpx_ctx *const *ctxs, int n_ctx, int in_fmt, int out_fmt, uint32_t samplerate,
                            uint32_t samplenum0, size_t slab_bytes, int slabs_per_ctx, dpx_stream **out)
{
    if (!ctxs || n_ctx < 1 || n_ctx > 64 || !out || !fmt_ok(in_fmt) || !fmt_ok(out_fmt) || slabs_per_ctx < 1 ||
        (long)slabs_per_ctx * n_ctx > 256)
        return fail(DPX_ERR_ARG, "bad argument");
    for (int i = 0; i < n_ctx; ++i)
        if (!ctxs[i]) return fail(DPX_ERR_ARG, "context %d is null", i);
    *out = nullptr;
    const size_t ibs = bytes_per_sample(in_fmt), obs = bytes_per_sample(out_fmt);
    slab_bytes = slab_bytes / 16 * 16;
    if (slab_bytes < 16) return fail(DPX_ERR_ARG, "slab_bytes must be at least 16");
    dpx_stream *s = new (std::nothrow) dpx_stream;
    if (!s) return fail(DPX_ERR_ARG, "out of host memory");
    s->ctx = ctxs[0];
    s->ctxs.assign(ctxs, ctxs + n_ctx);
    s->in_fmt = in_fmt;
    s->out_fmt = out_fmt;
    s->samplerate = samplerate;
    s->samplenum = samplenum0;
    s->slab_bytes = slab_bytes;
    s->slab_out = slab_bytes / ibs * obs;
    s->slabs.resize((size_t)slabs_per_ctx * (size_t)n_ctx);
    for (size_t k = 0; k < s->slabs.size(); ++k) {
        dpx_stream_slab &b = s->slabs[k];
        b.ctx = ctxs[k % (size_t)n_ctx];                 // consecutive slabs on consecutive GPUs: their copies and kernels overlap
        hipError_t e = hipSetDevice(b.ctx->device);
        // A slab's pinned buffers live on the NUMA node of ITS GPU (several GPUs only: one GPU's ring stays where its caller
        // runs): the pages are taken while the buffer is pinned, under this thread's policy.
        const int node = n_ctx > 1 ? gpu_numa_node(b.ctx->device) : -1;
        b.numa_node = node;
        if (node >= 0) prefer_numa_node(node);
        // portable: pinned for every device of the process, so that any slab can be handed to any GPU's DMA engines
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&b.h_in), slab_bytes, hipHostMallocPortable);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&b.h_out), s->slab_out + 16, hipHostMallocPortable);
        if (node >= 0) {
            if (e == hipSuccess) { memset(b.h_in, 0, slab_bytes); memset(b.h_out, 0, s->slab_out + 16); }   // first touch under the policy, in case pinning left any page untouched
            prefer_numa_node(-1);
        }
        if (e == hipSuccess) e = hipMalloc(&b.d_in, slab_bytes);
        if (e == hipSuccess) e = hipMalloc(&b.d_out, s->slab_out + 16);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b.done, hipEventDisableTiming);
        if (e != hipSuccess) {
            dpx_stream_destroy(s);
            return fail(DPX_ERR_HIP, "stream slab allocation failed: %s", hipGetErrorString(e));
        }
    }
    if (n_ctx > 1) {
        for (int i = 0; i < n_ctx; ++i) {
            s->workers.emplace_back(new dpx_stream::Worker);
            dpx_stream::Worker *w = s->workers.back().get();
            w->th = std::thread(slab_worker, s, w, s->slabs[(size_t)i].numa_node);
        }
    }
    *out = s;
    return DPX_OK;
}

int dpx_stream_create(dpx_ctx *ctx, int in_fmt, int out_fmt, uint32_t samplerate, uint32_t samplenum0,
                      size_t slab_bytes, int n_slabs, dpx_stream **out)
{
    if (!ctx) return fail(DPX_ERR_ARG, "bad argument");
    return dpx_stream_create_multi(&ctx, 1, in_fmt, out_fmt, samplerate, samplenum0, slab_bytes, n_slabs, out);
}

void dpx_stream_destroy(dpx_stream *s)
{
    if (!s) return;
    for (auto &w : s->workers) {                     // the enqueue threads finish what they were handed, then leave
        { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
    }
    for (dpx_stream_slab &b : s->slabs) {
        if (b.ctx) (void)hipSetDevice(b.ctx->device);
        if (b.stream) (void)hipStreamSynchronize(b.stream);
        if (b.h_in) (void)hipHostFree(b.h_in);
        if (b.h_out) (void)hipHostFree(b.h_out);
        if (b.d_in) (void)hipFree(b.d_in);
        if (b.d_out) (void)hipFree(b.d_out);
        release(b.dev);
        if (b.done) (void)hipEventDestroy(b.done);
        if (b.stream) (void)hipStreamDestroy(b.stream);
    }
    delete s;
}

int dpx_stream_acquire(dpx_stream *s, void **pinned_in, size_t *capacity_bytes)
{
    if (!s || !pinned_in) return fail(DPX_ERR_ARG, "bad argument");
    dpx_stream_slab &b = s->slabs[s->acq];
    if (b.state != 0) return fail(DPX_ERR_PLAN, "all %zu slabs are in use: call dpx_stream_next/release first", s->slabs.size());
    b.state = 1;
    s->acq = (s->acq + 1) % s->slabs.size();
    *pinned_in = b.h_in;
    if (capacity_bytes) *capacity_bytes = s->slab_bytes;
    return DPX_OK;
}

namespace {

// the device work of one submitted slab: plan image (unless the slab's resident one is reused), H2D, launch, D2H, event
int enqueue_slab(dpx_stream *s, dpx_stream_slab &b, double *upload_us, double *enqueue_us)
{
    using clk = std::chrono::steady_clock;
    auto us_since = [](clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); };
    DPX_ENTER(b.ctx);
    const clk::time_point t1 = clk::now();
    if (b.out_bytes == 0) {
        DPX_HIP(hipEventRecord(b.done, b.stream));
        return DPX_OK;
    }
    int rc;
    if (!b.job_reuse) {
        rc = materialize(b.ctx, b.plan, b.dev, b.job_fma, b.stream);
        if (rc != DPX_OK) return rc;
    }
    const clk::time_point t2 = clk::now();
    *upload_us += us_since(t1);
    DPX_HIP(hipMemcpyAsync(b.d_in, b.h_in, b.job_in_bytes, hipMemcpyHostToDevice, b.stream));
    rc = run_plan(b.plan, b.dev, b.d_in, s->in_fmt, b.d_out, s->out_fmt, b.job_fma, b.job_geom, b.stream);
    if (rc != DPX_OK) return rc;
    DPX_HIP(hipMemcpyAsync(b.h_out, b.d_out, b.out_bytes, hipMemcpyDeviceToHost, b.stream));
    DPX_HIP(hipEventRecord(b.done, b.stream));
    *enqueue_us += us_since(t2);
    return DPX_OK;
}

void slab_worker(dpx_stream *s, dpx_stream::Worker *w, int numa_node)
{
    // run where the GPU's pinned slabs live (sched_setaffinity to the node's CPUs: sysfs cpulist; silent on failure)
    if (numa_node >= 0) {
        char path[96];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", numa_node);
        if (FILE *f = fopen(path, "r")) {
            cpu_set_t set;
            CPU_ZERO(&set);
            int a = 0, b2 = 0, n = 0;
            char sep = 0;
            while (fscanf(f, "%d", &a) == 1) {
                b2 = a;
                if (fscanf(f, "%c", &sep) == 1 && sep == '-') { if (fscanf(f, "%d", &b2) != 1) b2 = a; if (fscanf(f, "%c", &sep) != 1) sep = 0; }
                for (int c = a; c <= b2 && c < CPU_SETSIZE; ++c) { CPU_SET(c, &set); ++n; }
                if (sep != ',') break;
            }
            fclose(f);
            if (n > 0) (void)sched_setaffinity(0, sizeof set, &set);
        }
    }
    for (;;) {
        size_t k;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->stop || !w->jobs.empty(); });
            if (w->jobs.empty()) return;
            k = w->jobs.front();
            w->jobs.pop_front();
        }
        dpx_stream_slab &b = s->slabs[k];
        double up = 0, en = 0;
        b.enq_rc = enqueue_slab(s, b, &up, &en);
        if (b.enq_rc != DPX_OK) b.enq_err = dpx_last_error();
        {
            std::lock_guard<std::mutex> lk(s->enq_mu);
            s->stats.upload_us += up;
            s->stats.enqueue_us += en;
            b.enq.store(2, std::memory_order_release);
        }
        s->enq_cv.notify_all();
    }
}

}  // namespace

int dpx_stream_submit(dpx_stream *s, size_t in_bytes, const dpx_segment *segs, size_t n_segs)
{
    if (!s || (n_segs && !segs)) return fail(DPX_ERR_ARG, "bad argument");
    dpx_stream_slab &b = s->slabs[s->head];
    if (b.state != 1) return fail(DPX_ERR_PLAN, "dpx_stream_submit without dpx_stream_acquire");
    const size_t ibs = bytes_per_sample(s->in_fmt), obs = bytes_per_sample(s->out_fmt);
    if (in_bytes > s->slab_bytes) return fail(DPX_ERR_CAPACITY, "%zu bytes exceed the slab (%zu)", in_bytes, s->slab_bytes);
    if (in_bytes % ibs != 0)
        return fail(DPX_ERR_BLOCK_LEN, "%zu bytes is not a whole number of samples", in_bytes);
    uint64_t total = 0;
    for (size_t i = 0; i < n_segs; ++i) total += segs[i].n_samples;
    if (total != in_bytes / ibs) return fail(DPX_ERR_PLAN, "segments hold %llu samples, the slab %zu",
                                             (unsigned long long)total, in_bytes / ibs);
    dpx_ctx *ctx = s->ctx;                 // planning state (period cache, tuning): the first context's
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    auto us_since = [](clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); };
    // A slab buffer remembers the plan it ran last and what it was made from: the same segments from the same counter
    // (const mode whenever the period divides the slab — the headline: every slab after the first round of the ring)
    // need neither planning nor a new device image, only the copies and the launch.
    uint32_t sn = s->samplenum;
    const dpx::LaunchGeom g = geometry(ctx);
    const SlabKey key = slab_key(ctx, g, segs, n_segs, s->samplerate, sn);
    bool reuse = total != 0 && b.have_key && b.key == key && b.key_segs.size() == n_segs;
    for (size_t i = 0; reuse && i < n_segs; ++i)             // field by field: the structs have padding
        reuse = b.key_segs[i].n_samples == segs[i].n_samples && memcmp(&b.key_segs[i].shift_hz, &segs[i].shift_hz, sizeof(float)) == 0;
    if (reuse) {
        sn = b.key_sn_after;
        ++s->stats.plans_reused;
    } else {
        b.have_key = false;
        b.plan = dpx::PlanResult();
        // the context remembers every ratio's period: a constant shift is scanned once per run, not once per slab
        append_segments(b.plan, segs, n_segs, s->samplerate, sn, ctx->variant, ctx->periods);
        dpx::finalize(b.plan, g.tile(), plan_choice(ctx), ctx->tuning);
        if (b.plan.error) return fail(DPX_ERR_PLAN, "%s", b.plan.error);
    }
    s->stats.plan_us += us_since(t0);
    b.out_bytes = (size_t)total * obs;
    b.job_in_bytes = in_bytes;
    b.job_reuse = reuse;
    b.job_geom = g;
    b.job_fma = ctx->fma;
    if (total && !reuse) {                  // (the key describes the plan; a failed enqueue is reported by dpx_stream_next)
        b.key = key;
        b.key_segs.assign(segs, segs + n_segs);
        b.key_sn_after = sn;
        b.have_key = true;
    }
    if (s->workers.empty()) {
        double up = 0, en = 0;
        const int rc = enqueue_slab(s, b, &up, &en);
        if (rc != DPX_OK) { b.have_key = false; return rc; }
        s->stats.upload_us += up;
        s->stats.enqueue_us += en;
        b.enq.store(0, std::memory_order_relaxed);
    } else {
        // the device work goes to the thread of this slab's GPU; dpx_stream_next waits for it, then for the event
        dpx_stream::Worker &w = *s->workers[s->head % s->workers.size()];
        b.enq.store(1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.jobs.push_back(s->head);
        }
        w.cv.notify_one();
    }
    ++s->stats.slabs;
    s->stats.total_us += us_since(t0);
    s->samplenum = sn;
    b.state = 2;
    s->head = (s->head + 1) % s->slabs.size();
    s->in_flight++;
    return DPX_OK;
}

int dpx_stream_pending(const dpx_stream *s, int *n)
{
    if (!s || !n) return fail(DPX_ERR_ARG, "bad argument");
    *n = s->in_flight;
    return DPX_OK;
}

int dpx_stream_next(dpx_stream *s, const void **pinned_out, size_t *out_bytes)
{
    if (!s || !pinned_out || !out_bytes) return fail(DPX_ERR_ARG, "bad argument");
    dpx_stream_slab &b = s->slabs[s->tail];
    if (b.state != 2) return fail(DPX_ERR_PLAN, "nothing in flight");
    if (b.enq.load(std::memory_order_acquire) != 0) {          // several GPUs: the slab's enqueue thread first
        std::unique_lock<std::mutex> lk(s->enq_mu);
        s->enq_cv.wait(lk, [&] { return b.enq.load(std::memory_order_acquire) == 2; });
        lk.unlock();
        b.enq.store(0, std::memory_order_relaxed);
        if (b.enq_rc != DPX_OK) {
            b.have_key = false;
            b.state = 3;                                       // the slab is handed out (empty) so that the ring keeps turning
            s->tail = (s->tail + 1) % s->slabs.size();
            *pinned_out = b.h_out;
            *out_bytes = 0;
            return fail(b.enq_rc, "%s", b.enq_err.c_str());
        }
        DPX_HIP(hipSetDevice(b.ctx->device));
    }
    DPX_HIP(hipEventSynchronize(b.done));
    b.state = 3;
    s->tail = (s->tail + 1) % s->slabs.size();
    *pinned_out = b.h_out;
    *out_bytes = b.out_bytes;
    return DPX_OK;
}

int dpx_stream_release(dpx_stream *s)
{
    if (!s) return fail(DPX_ERR_ARG, "bad argument");
    dpx_stream_slab &b = s->slabs[s->rel];
    if (b.state != 3) return fail(DPX_ERR_PLAN, "dpx_stream_release without dpx_stream_next");
    b.state = 0;
    s->rel = (s->rel + 1) % s->slabs.size();
    s->in_flight--;
    return DPX_OK;
}

int dpx_stream_get_stats(const dpx_stream *s, dpx_stream_stats *out)
{
    if (!s || !out) return fail(DPX_ERR_ARG, "bad argument");
    *out = s->stats;
    return DPX_OK;
}

int dpx_stream_samplenum(const dpx_stream *s, uint32_t *samplenum)
{
    if (!s || !samplenum) return fail(DPX_ERR_ARG, "bad argument");
    *samplenum = s->samplenum;
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
