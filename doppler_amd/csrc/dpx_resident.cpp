// dpx_resident.cpp — the block-per-call path: staging slots, the resident block kernel's protocol, dpx_shift_block_async / dpx_wait
// (one of the translation units behind include/doppler_hip*.h: see dpx_internal.h)
//
// The reference's loop hands ONE 8 KiB block per call to its operator (src/main.rs:113-118).  Instead of a launch per
// block, one kernel (dpx_kernels.hip, resident_block_kernel) stays on the GPU with a workgroup per staging slot and is
// handed blocks through doorbells in host-mapped memory (dpx_types.h, BlockCtl).  This file is the host half.
//
// Protocol, host side:
//   ring      payload and stretch list into the slot, then the control word (ticket | payload | - | ticket) with the first
//             ticket stored last, release.  The word proves itself to the reader (ctl_word_valid), so nothing depends on how
//             a 16-byte PCIe read is split.
//   serve     the kernel writes the output into the slot, fences, stores the ticket into `done`.
//   leave     on request (kDoorExit in every doorbell), by the idle clock (kResidentIdleTicks without a block), or when a
//             workgroup finds a ticket rung for ANOTHER instance of the kernel (format pair / libm build): its last store
//             per workgroup is state = parked.
//   A ticket is only ever served by the instance it was rung for: every slot records it; resident_drain() brings that
//   instance back if it left first, before anything else may run or be started on the context.
//   launches == stops + idle_exits + (running ? 1 : 0) at all times (dpx_resident_info): every launch ends in exactly one
//   of the two ways.
#include <stdlib.h>

#include <new>

#include "dpx_internal.h"

namespace dpx_api {

namespace {

constexpr uint64_t kResidentIdleTicks = 200000;       // 2 ms of the 100 MHz wall clock without a block: the kernel leaves
constexpr double kResidentTimeoutS = 5.0;             // a completion word that does not come: error, resident mode off

inline dpx::BlockCtl *slot_ctl(dpx_ctx::AsyncSlot &a) { return reinterpret_cast<dpx::BlockCtl *>(a.host + kSmallCtlOff); }
inline uint32_t load_acq(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void store_rel(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline void cpu_relax()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

// the control word of a slot: payload and the second ticket first, the first ticket last (release)
inline void ring(dpx::BlockCtl *c, uint32_t ticket, uint32_t n_samples, uint32_t n_segs, uint32_t legacy, uint32_t instance)
{
    c->payload = dpx::ctl_payload(ticket, n_samples, n_segs, legacy, instance);
    c->reserved = 0;
    c->doorbell2 = ticket;
    store_rel(&c->doorbell, ticket);
}

bool all_parked(dpx_ctx *ctx)
{
    for (auto &a : ctx->async_slots)
        if (a.host && load_acq(&slot_ctl(a)->state) != dpx::kResidentParked) return false;
    return true;
}

bool any_parked(dpx_ctx *ctx)
{
    for (auto &a : ctx->async_slots)
        if (a.host && load_acq(&slot_ctl(a)->state) == dpx::kResidentParked) return true;
    return false;
}

// the kernel is on its way out (asked, idle clock, foreign ticket): its workgroups go within microseconds of each other
int wait_parked(dpx_ctx *ctx, bool asked)
{
    const double t0 = mono_s();
    while (!all_parked(ctx)) {
        if (mono_s() - t0 > kResidentTimeoutS) {
            // it may only be queued behind other work and run later: told to leave, its slots out of use until it has (slot_usable)
            for (auto &s : ctx->async_slots)
                if (s.host) { ring(slot_ctl(s), dpx::kDoorExit, 0, 0, 0, 0); s.poisoned = true; }
            ctx->resident_on = false;
            // the device is handed back: other contexts must neither call into this one (it may be destroyed next) nor fail
            // their own launches on its account — theirs go per launch or start a kernel of their own, behind the stuck one
            // at worst.  This context keeps resident_running until its slots are seen parked (slot_usable).
            {
                dpx_ctx *me = ctx;
                ctx->dev->resident_owner.compare_exchange_strong(me, nullptr);
            }
            return fail(DPX_ERR_HIP, "the resident block kernel does not leave");
        }
        cpu_relax();
    }
    DPX_HIP(hipStreamSynchronize(ctx->rstream));
    ctx->resident_running.store(false, std::memory_order_release);
    dpx_ctx *me = ctx;
    ctx->dev->resident_owner.compare_exchange_strong(me, nullptr);
    if (asked) ++ctx->resident_stops; else ++ctx->resident_idle_exits;
    return DPX_OK;
}

// start the instance (in_fmt, out_fmt, fma); no kernel of this context may be running — one of ANOTHER context of the
// device is asked to leave first (dpx_internal.h, DeviceState: one resident kernel per device and process)
int launch(dpx_ctx *ctx, int in_fmt, int out_fmt, bool fma)
{
    if (dpx_ctx *other = ctx->dev->resident_owner.load(std::memory_order_acquire)) {
        if (other != ctx) {
            const int rc = resident_stop(other);
            if (rc != DPX_OK) return rc;
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
        // a stop request of the past is not one for this launch: the doorbell shows the last ticket served again
        if (c->doorbell == dpx::kDoorExit) ring(c, c->done, 0, 0, 0, 0);
        store_rel(&c->state, dpx::kResidentRunning);
        ra.ctl[k] = reinterpret_cast<dpx::BlockCtl *>(a.dev + kSmallCtlOff);
        ra.in[k] = reinterpret_cast<const uint8_t *>(a.dev + kSmallInOff);
        ra.out[k] = reinterpret_cast<uint8_t *>(a.dev + kSmallOutOff);
        ra.segs[k] = reinterpret_cast<const dpx::DevSeg *>(a.dev + kSmallPlanOff);
    }
    ra.shared = ctx->rshared;
    ra.idle_ticks = kResidentIdleTicks;
    const int rc = dpx::launch_resident_block(ra, in_fmt, out_fmt, fma, ctx->rstream);
    if (rc != DPX_OK) {
        for (auto &a : ctx->async_slots) slot_ctl(a)->state = dpx::kResidentParked;
        return fail(rc, "resident block kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    ctx->resident_running.store(true, std::memory_order_release);
    ctx->dev->resident_owner.store(ctx, std::memory_order_release);
    ctx->resident_in = in_fmt;
    ctx->resident_out = out_fmt;
    ctx->resident_fma = fma;
    ++ctx->resident_launches;
    return DPX_OK;
}

inline bool unserved(dpx_ctx::AsyncSlot &a)
{
    return a.host && a.resident && a.seq != 0 && load_acq(&slot_ctl(a)->done) != a.seq;
}

// Every ticket that has been rung is served when this returns DPX_OK — by the instance it was rung for, which is started
// again if it left first (idle clock, or it met a ticket of another instance).
int resident_drain(dpx_ctx *ctx)
{
    const double t0 = mono_s();
    for (;;) {
        dpx_ctx::AsyncSlot *pending = nullptr;
        for (auto &a : ctx->async_slots)
            if (unserved(a)) { pending = &a; break; }
        if (!pending) return DPX_OK;
        const bool running = ctx->resident_running.load(std::memory_order_acquire);
        const bool right = running && ctx->resident_in == pending->in_fmt && ctx->resident_out == pending->out_fmt && ctx->resident_fma == pending->fma;
        if (!running || any_parked(ctx) || !right) {
            if (running) {
                // the kernel is leaving, or is another instance (which leaves when it sees the ticket): wait it out
                if (!right && !any_parked(ctx))
                    for (auto &a : ctx->async_slots)
                        if (a.host && !unserved(a)) ring(slot_ctl(a), dpx::kDoorExit, 0, 0, 0, 0);
                const int rc = wait_parked(ctx, !right);
                if (rc != DPX_OK) return rc;
            }
            const int rc = launch(ctx, pending->in_fmt, pending->out_fmt, pending->fma);
            if (rc != DPX_OK) return rc;
        }
        if (mono_s() - t0 > kResidentTimeoutS) {
            ctx->resident_on = false;
            return fail(DPX_ERR_HIP, "the resident block kernel did not finish ticket %u", pending->seq);
        }
        cpu_relax();
    }
}

// A resident kernel of the instance (in_fmt, out_fmt, ctx->fma) is polling every slot's doorbell when this returns.
int resident_ensure(dpx_ctx *ctx, int in_fmt, int out_fmt)
{
    if (ctx->resident_running.load(std::memory_order_acquire)) {
        const bool same = ctx->resident_in == in_fmt && ctx->resident_out == out_fmt && ctx->resident_fma == ctx->fma;
        if (same && !any_parked(ctx)) return DPX_OK;
        if (!same) {
            const int rc = resident_stop(ctx);        // serves what is rung for the old instance first
            if (rc != DPX_OK) return rc;
        } else {
            const int rc = wait_parked(ctx, false);   // the idle clock took it
            if (rc != DPX_OK) return rc;
        }
    }
    return launch(ctx, in_fmt, out_fmt, ctx->fma);
}

}  // namespace

int alloc_slot(dpx_ctx::AsyncSlot &a)
{
    if (a.host) return DPX_OK;
    void *h = nullptr, *d = nullptr;
    // mapped and coherent (fine-grained: uncached on the device) — the doorbell protocol needs every store to be visible
    // to the other side without a cache flush; stated rather than implied by hipHostMallocMapped
    DPX_HIP(hipHostMalloc(&h, kSlotBytes, hipHostMallocMapped | hipHostMallocCoherent));
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

// a slot a resident kernel stopped answering on may still be written by it: usable again once every workgroup is parked
bool slot_usable(dpx_ctx *ctx, dpx_ctx::AsyncSlot &a)
{
    if (!a.poisoned) return true;
    if (!all_parked(ctx)) return false;
    for (auto &s : ctx->async_slots) s.poisoned = false;
    ctx->resident_running.store(false, std::memory_order_release);
    dpx_ctx *me = ctx;
    ctx->dev->resident_owner.compare_exchange_strong(me, nullptr);
    return true;
}

// Ask the resident kernel to leave and wait until it has: before any launch of this context's own (a resident kernel holds
// its hardware queue; another stream's launch that shares the queue would wait for it), before the context goes away, and
// when another instance is needed.  Blocks already rung are served first.  Costs one load when no kernel is running.
int resident_stop(dpx_ctx *ctx)
{
    if (!ctx->resident_running.load(std::memory_order_acquire)) return DPX_OK;
    int rc = resident_drain(ctx);
    if (rc != DPX_OK) return rc;
    if (!ctx->resident_running.load(std::memory_order_acquire)) return DPX_OK;
    for (auto &a : ctx->async_slots)
        if (a.host) ring(slot_ctl(a), dpx::kDoorExit, 0, 0, 0, 0);
    return wait_parked(ctx, true);
}

int resident_stop_device(dpx_ctx *ctx)
{
    dpx_ctx *owner = ctx->dev->resident_owner.load(std::memory_order_acquire);
    return owner ? resident_stop(owner) : DPX_OK;
}

}  // namespace dpx_api

using namespace dpx_api;

extern "C" {

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
    std::lock_guard<std::recursive_mutex> lock(ctx->dev->mu);
    DPX_HIP(hipSetDevice(ctx->device));
    const uint32_t seq = ctx->async_next_seq;
    dpx_ctx::AsyncSlot &a = ctx->async_slots[seq % dpx_ctx::kAsyncSlots];
    if (a.seq != 0) return fail(DPX_ERR_PLAN, "%d blocks are in flight: dpx_wait for ticket %u first", dpx_ctx::kAsyncSlots, a.seq);
    int rc = alloc_slot(a);
    if (rc != DPX_OK) return rc;
    if (!slot_usable(ctx, a)) return fail(DPX_ERR_HIP, "a resident block kernel that stopped answering still holds this context's staging slots");
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
        rc = resident_stop_device(ctx);
        if (rc != DPX_OK) return rc;
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
        rc = resident_ensure(ctx, in_fmt, out_fmt);
        if (rc == DPX_OK) {
            a.in_fmt = in_fmt;
            a.out_fmt = out_fmt;
            a.fma = ctx->fma;
            ring(c, seq, (uint32_t)n, (uint32_t)plan.segs.size(),
                 (ctx->i16_cast == DPX_CAST_LEGACY_X86 && out_fmt == DPX_FMT_I16) ? 1u : 0u, dpx::ctl_instance(in_fmt, out_fmt, ctx->fma));
            a.resident = true;
            ++ctx->resident_blocks;
            return issue(sn);
        }
        if (ctx->resident_on) return rc;              // (a kernel that does not answer turns the mode off: the launch path below)
    }
    // ---- one launch per block (round 3's path): periods the resident kernel's slot cannot hold, or resident mode off
    rc = resident_stop_device(ctx);
    if (rc != DPX_OK) return rc;
    if (!slot_usable(ctx, a)) return fail(DPX_ERR_HIP, "a resident block kernel that stopped answering still holds this context's staging slots");
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
    std::lock_guard<std::recursive_mutex> lock(ctx->dev->mu);
    dpx_ctx::AsyncSlot &a = ctx->async_slots[ticket % dpx_ctx::kAsyncSlots];
    if (a.seq != ticket) return fail(DPX_ERR_ARG, "ticket %u is not in flight", ticket);
    if (a.out_bytes > out_cap || (!out && a.out_bytes))
        return fail(DPX_ERR_CAPACITY, "output needs %zu bytes, capacity %zu", a.out_bytes, out_cap);
    if (a.resident) {
        // the block's completion word in host memory; a kernel that left meanwhile (idle clock, a ticket of another instance)
        // is started again — the instance this ticket was rung for — and finds the doorbell rung
        dpx::BlockCtl *c = slot_ctl(a);
        const double t0 = mono_s();
        uint32_t spins = 0;
        while (load_acq(&c->done) != ticket) {
            if ((++spins & 63u) == 0) {
                if (load_acq(&c->state) == dpx::kResidentParked && load_acq(&c->done) != ticket) {
                    DPX_HIP(hipSetDevice(ctx->device));
                    int rc = DPX_OK;
                    if (ctx->resident_running.load(std::memory_order_acquire)) rc = wait_parked(ctx, false);
                    if (rc == DPX_OK && load_acq(&c->done) != ticket) rc = launch(ctx, a.in_fmt, a.out_fmt, a.fma);
                    if (rc != DPX_OK) { a.seq = 0; return rc; }
                }
                if (mono_s() - t0 > kResidentTimeoutS) {
                    // The kernel may only be queued behind other work and run later: it must not find this ticket then, and the
                    // slots stay out of use until every workgroup has been seen parked (slot_usable).
                    for (auto &s : ctx->async_slots)
                        if (s.host) { ring(slot_ctl(s), dpx::kDoorExit, 0, 0, 0, 0); s.poisoned = true; if (s.resident) s.seq = 0; }
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
    std::lock_guard<std::recursive_mutex> lock(ctx->dev->mu);
    if (!on) {
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

int dpx_resident_info(dpx_ctx *ctx, dpx_resident_counters *out)
{
    if (!ctx || !out) return fail(DPX_ERR_ARG, "bad argument");
    std::lock_guard<std::recursive_mutex> lock(ctx->dev->mu);
    memset(out, 0, sizeof *out);
    out->launches = ctx->resident_launches;
    out->blocks = ctx->resident_blocks;
    out->stops = ctx->resident_stops;
    out->idle_exits = ctx->resident_idle_exits;
    out->running = ctx->resident_running.load(std::memory_order_acquire) ? 1u : 0u;
    for (auto &a : ctx->async_slots) {
        if (!a.host) continue;
        if (a.seq != 0) ++out->tickets_in_flight;
        if (load_acq(&slot_ctl(a)->state) == dpx::kResidentParked) ++out->slots_parked;
    }
    return DPX_OK;
}

}  // extern "C"
