// dpx_internal.h — what the translation units behind include/doppler_hip*.h share (not installed, not part of the ABI).
//
//   dpx_context.cpp    context, tuning knobs, plan images on the device (materialize / run_plan), memory helpers
//   dpx_resident.cpp   the block-per-call path: staging slots, the resident block kernel's protocol, dpx_shift_block_async / dpx_wait
//   dpx_operators.cpp  the host-pointer operators of doppler::dsp and the host-only arithmetic (counter algebra, schedule, orbit)
//   dpx_plans.cpp      plans over device buffers (dpx_plan_* / dpx_run_device) and the planner's self-checks
//   dpx_stream.cpp     the slab ring (dpx_stream_*), one GPU or several
// There is deliberately no CPU implementation of any entry point in any of them: if the GPU or the kernels are
// unavailable the calls fail with an error code.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>

#include "../../include/doppler_hip.h"
#include "../../include/doppler_hip_debug.h"
#include "../../include/doppler_hip_host.h"
#include "dpx_planner.h"
#include "dpx_types.h"

struct dpx_ctx;

namespace dpx_api {

// sets the thread-local message dpx_last_error() returns; returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3), visibility("hidden")));

#define DPX_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return dpx_api::fail(DPX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));     \
    } while (0)

inline size_t bytes_per_sample(int fmt) { return fmt == DPX_FMT_I16 ? 4 : 8; }
inline bool fmt_ok(int fmt) { return fmt == DPX_FMT_I16 || fmt == DPX_FMT_F32; }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline double mono_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// device image of a plan: stretch table | hint table | span descriptors | leftover ranges | corrector-table pool, one allocation
struct DevPlan {
    void *buf = nullptr;
    size_t cap = 0;
    dpx::DevSeg *segs = nullptr;
    uint32_t *hint = nullptr;
    dpx::WalkSeg *walk = nullptr;   // one descriptor per 2^kWalkHintShift workgroups of the span launch
    dpx::LeftRange *left = nullptr;
    uint32_t *left_hint = nullptr;
    void *lut = nullptr;
    std::vector<char> image;       // host copy of everything before the tables, source of the one upload
};

// layout of a block-per-call staging buffer (pinned, device-mapped): input | output | stretch list (+ hint table) | control lines
constexpr size_t kSmallCallBytes = 64 << 10;      // per side; larger calls go through device staging buffers
constexpr size_t kSmallPlanBytes = 16 << 10;
constexpr size_t kSmallInOff = 0, kSmallOutOff = kSmallCallBytes, kSmallPlanOff = 2 * kSmallCallBytes;
constexpr size_t kSmallCtlOff = 2 * kSmallCallBytes + kSmallPlanBytes;      // dpx::BlockCtl of an asynchronous slot
constexpr size_t kSlotBytes = kSmallCtlOff + 256;

// Per device, process-wide: ONE lock for every entry point that touches a context of that device, and the context whose
// resident block kernel is running there, if any.  A resident kernel holds the hardware queue its stream was mapped to, and
// HIP maps streams to a handful of hardware queues as it pleases: two contexts of one process whose resident kernels land
// on the same queue take turns at the pace of the 2 ms idle clock (measured: 400 + 400 interleaved blocks in 1.6 s instead of
// 14 ms), and ANY launch of another context can queue behind a resident kernel that is not its own.  So there is at most one
// resident kernel per device and process, and every launch of the library on that device asks it to leave first, whichever
// context started it (contexts that alternate block by block hand it over at ~25 us per block: bounded, and never wrong).
struct DeviceState {
    std::recursive_mutex mu;
    std::atomic<dpx_ctx *> resident_owner{nullptr};
};
DeviceState &device_state(int device);

}  // namespace dpx_api

struct dpx_ctx {
    int device = -1;
    int n_cu = 0;
    bool fma = true;          // libm variant whose sincosf the kernels reproduce
    int block = 128;          // tile kernel: lanes per workgroup (128 or 256)
    int vecs = 2;             // tile kernel: 4-sample groups per lane (1 or 2)
    bool geom_auto = true;    // until dpx_set_tuning names a geometry: chosen per launch (run_plan)
    int variant = 0;
    int choice = dpx::kChooseAuto;   // which kernels finalize() may use (dpx_set_tuning)
    int i16_cast = DPX_CAST_SATURATE;   // meaning of `as i16` (dpx_set_i16_cast)
    dpx::PlanTuning tuning;          // kernel-shape knobs (dpx_set_options)
    dpx::PeriodCache periods;        // period per ratio seen so far (one producer thread plans at a time)
    hipStream_t stream = nullptr;   // internal stream of the host-pointer entry points
    void *stage_in = nullptr;
    void *stage_out = nullptr;
    size_t stage_in_cap = 0, stage_out_cap = 0;
    dpx_api::DevPlan *scratch = nullptr;   // device side of the host-pointer operators' plans
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
        int in_fmt = 0, out_fmt = 0; // ... and the kernel instance it was rung for (a ticket is only ever served by that one)
        bool fma = true;
        bool poisoned = false;       // a resident kernel that stopped answering may still write this slot: unusable until it is seen parked
    } async_slots[kAsyncSlots];
    uint32_t async_next_seq = 1;
    // the resident block kernel (dpx_types.h, BlockCtl): one workgroup per slot, launched once, polling the slots' doorbells
    bool resident_on = true;         // DPX_RESIDENT=0 or dpx_set_resident(ctx, 0): every block is a launch, as in round 3
    std::atomic<bool> resident_running{false};   // host's view: a kernel has been launched and not yet seen parked
    int resident_in = -1, resident_out = -1;
    bool resident_fma = true;
    hipStream_t rstream = nullptr;
    dpx::ResidentShared *rshared = nullptr;
    uint64_t resident_launches = 0, resident_blocks = 0;
    uint64_t resident_stops = 0, resident_idle_exits = 0;   // how the launches ended: asked to leave / found parked (idle clock)
    // A context is one caller's at a time (include/doppler_hip.h) — but a context lent to a multi-GPU stream is also used
    // by that stream's enqueue thread, and the resident kernel is a matter of the whole device (DeviceState).  Every entry
    // point that touches the stream / staging / resident state holds the device's lock (recursive: dpx_shift_block ->
    // dpx_shift_block_async -> the launch path), so two callers never interleave inside one.
    dpx_api::DeviceState *dev = nullptr;
};
static_assert(dpx_ctx::kAsyncSlots == dpx::kResidentSlots, "one resident workgroup per staging slot");

struct dpx_plan {
    dpx_ctx *ctx = nullptr;
    dpx::PlanResult host;
    dpx::LaunchGeom geom;
    bool fma = true;
    dpx_api::DevPlan dev;
    double t_append_us = 0, t_finalize_us = 0, t_upload_us = 0;   // where dpx_plan_segments spent its time (dpx_plan_timing)
};

namespace dpx_api {

// ---- dpx_context.cpp
int ensure_stage(dpx_ctx *ctx, size_t in_bytes, size_t out_bytes);
dpx::LaunchGeom geometry(const dpx_ctx *ctx);
int choice_of(int variant);                 // dpx_set_tuning variants 4..6 restrict the kernels a plan may use
dpx::PlanTuning tuning_of(const dpx_options *o);
// the stretch list of a segment list (counter carried from segment to segment), periods scanned in parallel first
void append_segments(dpx::PlanResult &plan, const dpx_segment *segs, size_t n_segs, uint32_t samplerate, uint32_t &sn,
                     int variant, dpx::PeriodCache &cache);
// upload stretch + hint tables and fill the corrector tables (async on `st`); `plan` must have been finalize()d
int materialize(dpx_ctx *ctx, const dpx::PlanResult &plan, DevPlan &dev, bool fma, hipStream_t st);
// every launch of a finalized plan, asynchronously on `st`
int run_plan(const dpx::PlanResult &plan, const DevPlan &dev, const void *d_in, int in_fmt, void *d_out,
             int out_fmt, bool fma, const dpx::LaunchGeom &g_in, void *st);
void release(DevPlan &dev);

// ---- dpx_resident.cpp
int alloc_slot(dpx_ctx::AsyncSlot &a);
int resident_stop(dpx_ctx *ctx);            // this context's resident kernel has served what was rung and has left when this returns DPX_OK
int resident_stop_device(dpx_ctx *ctx);     // ... whichever context of ctx's device owns the running kernel
bool slot_usable(dpx_ctx *ctx, dpx_ctx::AsyncSlot &a);

// ---- dpx_operators.cpp
int run_host(dpx_ctx *ctx, const void *in, size_t n, int in_fmt, void *out, int out_fmt,
             uint32_t *samplenum, float shift_hz, uint32_t samplerate);

}  // namespace dpx_api

// Every entry point that launches or synchronises on a context's behalf passes through here first: the context's lock
// for the rest of the function, its device current, and any resident block kernel on that device (which holds a hardware queue) gone.
#define DPX_ENTER(ctx)                                                             \
    std::lock_guard<std::recursive_mutex> dpx_ctx_lock_((ctx)->dev->mu);           \
    do {                                                                           \
        DPX_HIP(hipSetDevice((ctx)->device));                                      \
        if ((ctx)->dev->resident_owner.load(std::memory_order_acquire)) {          \
            const int rc_enter_ = dpx_api::resident_stop_device(ctx);              \
            if (rc_enter_ != DPX_OK) return rc_enter_;                             \
        }                                                                          \
    } while (0)
