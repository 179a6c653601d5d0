// dpx_stream.cpp — the slab ring: streaming from host memory, on one GPU or several
// (one of the translation units behind include/doppler_hip*.h: see dpx_internal.h)
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <memory>
#include <new>
#include <string>
#include <thread>

#include "dpx_internal.h"

using namespace dpx_api;

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
    void *m_in = nullptr, *m_out = nullptr;   // the pinned buffers as the GPU addresses them (hipHostGetDevicePointer): the direct path's kernel arguments
    void *d_in = nullptr, *d_out = nullptr;   // HBM staging of the copy-engine path (allocated only for the sides that are staged)
    void *g_out = nullptr;                    // RCCL gather: where the slab's output lands on the ring's first GPU (slabs of other GPUs only)
    hipEvent_t ev_gather = nullptr;           // ... and when it has
    hipStream_t stream = nullptr;             // the slab's launches: its own stream where the kernel itself crosses PCIe (launches of several slabs
                                              // overlap) and on the per-slab form of the staged path; the GPU's `run` stream on the staged path
    bool owns_stream = false;
    hipEvent_t done = nullptr;
    hipEvent_t ev_up = nullptr, ev_run = nullptr;   // input has arrived in HBM / the launches are done: what links the three streams a staged slab crosses
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
    // How a slab crosses PCIe (dpx_stream_options.path).  DIRECT: the fused kernel loads from the pinned input slab and stores
    // to the pinned output slab themselves — the samples cross the link once each way and never rest in HBM, reads and writes
    // of ONE launch keep both directions of the link busy.  STAGED: copy engine H2D -> kernel HBM to HBM -> copy engine D2H on
    // the slab's stream (rounds 2-5).  The two mixed forms stage one side only.  profiles/r06_ring.md has the same-process A/B.
    uint32_t path = DPX_STREAM_PATH_DIRECT;
    bool copy_only = false;      // calibration: the same slabs, the same path, no arithmetic (what the link gives the ring)
    bool in_direct() const { return path == DPX_STREAM_PATH_DIRECT || path == DPX_STREAM_PATH_DIRECT_IN; }
    bool out_direct() const { return path == DPX_STREAM_PATH_DIRECT || path == DPX_STREAM_PATH_DIRECT_OUT; }
    bool per_slab() const { return path == DPX_STREAM_PATH_STAGED_PER_SLAB; }
    // One stream per direction and GPU for the copy engines: every H2D of a GPU's slabs queues on `up`, every D2H on `down`,
    // the launches stay on the slabs' own streams, events in between.  Measured (tools/pcie_probe.hip, profiles/r06_ring.md):
    // ONE H2D stream against ONE D2H stream moves 56 + 48 GB/s; H2D -> kernel -> D2H on a stream per slab (rounds 2-5) takes
    // turns at the link (28 GB/s each way: 1 / (1/57 + 1/51)) however many slabs are in flight.
    // The D2H copies are PACED: a slab's copy is handed to the runtime only when the previous one of its GPU has finished.
    // The runtime picks the copy engine of a D2H when it is submitted — the lowest engine that is idle at that moment
    // (AMD_LOG_LEVEL=4: H2D always engine 0, D2H engines 1, 2, 3, ... as earlier ones still wait for their kernels) — and
    // several engines copying D2H at once share the link badly: a ring of 4 (6) slabs whose D2H were all queued up front
    // moved 27 (18-22) GB/s each way, the same ring paced 46-48.  dpx_stream_submit and dpx_stream_next both pump the queue;
    // next(k) always can (the slab before k on its GPU has been handed out, so its copy is done).
    struct Lane {
        hipStream_t up = nullptr, run = nullptr, down = nullptr;
        std::mutex mu;                     // guards down_q / last_down (producer and consumer threads both pump)
        std::deque<size_t> down_q;         // slabs whose launches are enqueued and whose D2H is not yet
        long last_down = -1;               // slab of the newest D2H handed to the runtime
        std::vector<hipStream_t> parked;   // streams that shared a hardware queue with another of the three: kept (idle) so that
                                           // their replacements are dealt a different queue, destroyed with the ring
        int probes = 0;                    // rounds of separate_lane_streams (dpx_stream_describe: 0 = not probed)
        bool shared_queue = false;         // ... and whether the last round still saw two of the streams wait for each other
    };
    std::vector<std::unique_ptr<Lane>> lanes;     // one per context
    bool paced = true;                     // dpx_stream_options.path | DPX_STREAM_UNPACED: every D2H queued at submit time (A/B)
    // dpx_stream_options.gather == DPX_STREAM_GATHER_RCCL (what BASELINE.json's north_star names: "RCCL over xGMI only for
    // ordered gather back to stdout"): a slab of GPU g != 0 does not leave through g's own PCIe link; its output is sent
    // over xGMI into a buffer on the ring's first GPU (one ncclSend / ncclRecv pair in one group, one communicator per GPU
    // of this ONE process: ncclCommInitAll) and leaves from there, paced like that GPU's own slabs.  Input still goes to
    // every GPU over its own link.  Every output byte then crosses GPU 0's link: the default (per-GPU D2H) is N times wider
    // for a host consumer — README.md says so; this is the form the north_star asks for, behind the same ordering.
    bool gather_rccl = false;
    bool gather_self = false;              // DPX_STREAM_GATHER_SELF (tests on one GPU): GPU 0's own slabs go through a send/recv to itself
    std::vector<void *> comms;             // ncclComm_t per context
    hipStream_t gather_stream = nullptr;   // on GPU 0: the receives, in the order the groups were issued
    std::mutex gather_mu;                  // one group at a time touches GPU 0's communicator (the enqueue threads of several GPUs)
    uint64_t gathered_slabs = 0, gathered_bytes = 0;       // through ncclSend / ncclRecv so far (dpx_stream_describe)
    bool gathered(size_t k) const { return gather_rccl && (k % ctxs.size() != 0 || gather_self); }
    size_t down_lane(size_t k) const { return gathered(k) ? 0 : k % lanes.size(); }       // the GPU whose link a slab's output leaves through
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
    mutable std::mutex enq_mu;     // guards stats' upload / enqueue parts and wakes dpx_stream_next
    std::condition_variable enq_cv;
};

namespace {
constexpr size_t kDirectBelow = 1u << 20, kStagedFrom = 4u << 20;     // slab sizes at which the default path changes (dpx_stream_create_opts)
void slab_worker(dpx_stream *s, dpx_stream::Worker *w, int numa_node);
int pump_down(dpx_stream *s, dpx_stream::Lane &lane, long upto);
void separate_lane_streams(dpx_stream *s, size_t lane_index);
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

// librccl, loaded when a ring asks for the RCCL gather (not a link-time dependency: a one-GPU user never needs it).
// The handful of entry points used, with the signatures of rccl.h (ncclComm_t is an opaque pointer, ncclUint8 == 1).
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int datatype, int peer, void *comm, hipStream_t stream) = nullptr;
    int (*Recv)(void *buf, size_t count, int datatype, int peer, void *comm, hipStream_t stream) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int kNcclUint8 = 1;

const Rccl &rccl()
{
    static const Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.lib, "ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.lib, "ncclCommDestroy"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.lib, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.lib, "ncclGroupEnd"));
        x.Send = reinterpret_cast<decltype(x.Send)>(dlsym(x.lib, "ncclSend"));
        x.Recv = reinterpret_cast<decltype(x.Recv)>(dlsym(x.lib, "ncclRecv"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.lib, "ncclGetErrorString"));
        x.ok = x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd && x.Send && x.Recv && x.GetErrorString;
        return x;
    }();
    return r;
}

#define DPX_NCCL(call)                                                                                     \
    do {                                                                                                   \
        const int rc_nccl_ = (call);                                                                       \
        if (rc_nccl_ != 0) return dpx_api::fail(DPX_ERR_HIP, "%s failed: %s", #call, rccl().GetErrorString(rc_nccl_)); \
    } while (0)

}  // namespace

extern "C" {

int dpx_stream_create_opts(dpx_ctx *const *ctxs, int n_ctx, int in_fmt, int out_fmt, uint32_t samplerate,
                           uint32_t samplenum0, size_t slab_bytes, int slabs_per_ctx, const dpx_stream_options *opt,
                           dpx_stream **out)
{
    if (!ctxs || n_ctx < 1 || n_ctx > 64 || !out || !fmt_ok(in_fmt) || !fmt_ok(out_fmt) || slabs_per_ctx < 1 ||
        (long)slabs_per_ctx * n_ctx > 256)
        return fail(DPX_ERR_ARG, "bad argument");
    dpx_stream_options o = {};
    if (opt) o = *opt;
    else if (const char *e = getenv("DPX_STREAM_PATH")) o.path = (uint32_t)atoi(e);      // A/B of the shipped command without a rebuild
    if ((o.path & 0xffu) > DPX_STREAM_PATH_STAGED_PER_SLAB || (o.path & ~(0xffu | DPX_STREAM_COPY_ONLY | DPX_STREAM_UNPACED | DPX_STREAM_NO_PROBE)))
        return fail(DPX_ERR_ARG, "unknown stream path %u", o.path);
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
    // The default follows the slab size (same-process A/B on two boxes, i16 -> i16, GB/s of input; profiles/r06_ring.md):
    //   slab       8 KiB  64 KiB  256 KiB  2 MiB  4 MiB  8 MiB  16 MiB  64 MiB
    //   DIRECT      0.8    6.6     22.2    29.6   30.5   32.3   36.6    41.9     one launch, no copy engine: latency wins
    //   DIRECT_OUT  0.5    3.2     11.1    33.6   39.0   38.9   41.4    44.0     engine H2D against the kernel's own stores
    //   STAGED      0.3    2.3      7.9    33.2   38.7   43.1   45.8    47.2     engine H2D against engine D2H, paced
    s->path = (o.path & 0xffu) != DPX_STREAM_PATH_DEFAULT ? (o.path & 0xffu)
              : slab_bytes < kDirectBelow ? (uint32_t)DPX_STREAM_PATH_DIRECT
              : slab_bytes < kStagedFrom ? (uint32_t)DPX_STREAM_PATH_DIRECT_OUT : (uint32_t)DPX_STREAM_PATH_STAGED;
    s->copy_only = (o.path & DPX_STREAM_COPY_ONLY) != 0;
    s->slabs.resize((size_t)slabs_per_ctx * (size_t)n_ctx);
    for (int i = 0; i < n_ctx; ++i) s->lanes.emplace_back(new dpx_stream::Lane);
    s->paced = (o.path & DPX_STREAM_UNPACED) == 0;
    if (!opt) if (const char *e = getenv("DPX_STREAM_GATHER")) o.gather = strcmp(e, "rccl") == 0 ? DPX_STREAM_GATHER_RCCL : (uint32_t)atoi(e);
    if ((o.gather & 0xffu) > DPX_STREAM_GATHER_RCCL || (o.gather & ~(0xffu | DPX_STREAM_GATHER_SELF))) {
        dpx_stream_destroy(s);
        return fail(DPX_ERR_ARG, "unknown gather mode %u", o.gather);
    }
    s->gather_rccl = (o.gather & 0xffu) == DPX_STREAM_GATHER_RCCL;
    s->gather_self = s->gather_rccl && (o.gather & DPX_STREAM_GATHER_SELF) != 0;
    if (s->gather_rccl) {
        if ((o.path & 0xffu) != DPX_STREAM_PATH_DEFAULT && (o.path & 0xffu) != DPX_STREAM_PATH_STAGED) {
            dpx_stream_destroy(s);
            return fail(DPX_ERR_ARG, "the RCCL gather runs on the staged path");
        }
        s->path = DPX_STREAM_PATH_STAGED;                 // outputs rest in HBM before they travel: whatever the slab size
        for (int i = 0; i < n_ctx; ++i)
            for (int j = 0; j < i; ++j)
                if (ctxs[i]->device == ctxs[j]->device) {
                    dpx_stream_destroy(s);
                    return fail(DPX_ERR_ARG, "the RCCL gather needs distinct devices (device %d is listed twice): one communicator rank per GPU", ctxs[i]->device);
                }
        if (!rccl().ok) {
            dpx_stream_destroy(s);
            return fail(DPX_ERR_HIP, "librccl.so.1 could not be loaded or lacks an entry point: the RCCL gather is unavailable, the default per-GPU D2H is not");
        }
    }
    for (size_t k = 0; k < s->slabs.size(); ++k) {
        dpx_stream_slab &b = s->slabs[k];
        b.ctx = ctxs[k % (size_t)n_ctx];                 // consecutive slabs on consecutive GPUs: their copies and kernels overlap
        hipError_t e = hipSetDevice(b.ctx->device);
        // A slab's pinned buffers live on the NUMA node of ITS GPU: the pages are taken while the buffer is pinned, under this
        // thread's policy.  One GPU too (round 6): a ring whose slabs landed on the other socket — wherever the creating thread
        // happened to run — moved 26 GB/s each way instead of 47, one ring instance in ten on a two-socket box.
        const int node = gpu_numa_node(b.ctx->device);
        b.numa_node = node;
        if (node >= 0) prefer_numa_node(node);
        // portable: pinned for every device of the process, so that any slab can be handed to any GPU's DMA engines
        // (mapped: the direct path's kernels address them; extra flags — hipHostMallocNonCoherent, WriteCombined, NumaUser —
        // only through dpx_stream_options, for the A/B of profiles/r06_ring.md)
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&b.h_in), slab_bytes, hipHostMallocPortable | hipHostMallocMapped | o.in_host_flags);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&b.h_out), s->slab_out + 16, hipHostMallocPortable | hipHostMallocMapped | o.out_host_flags);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&b.m_in, b.h_in, 0);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&b.m_out, b.h_out, 0);
        if (node >= 0) {
            if (e == hipSuccess) { memset(b.h_in, 0, slab_bytes); memset(b.h_out, 0, s->slab_out + 16); }   // first touch under the policy, in case pinning left any page untouched
            prefer_numa_node(-1);
        }
        if (e == hipSuccess && !s->in_direct()) e = hipMalloc(&b.d_in, slab_bytes);
        if (e == hipSuccess && !s->out_direct()) e = hipMalloc(&b.d_out, s->slab_out + 16);
        if (e == hipSuccess && k < (size_t)n_ctx) {
            // The first slab of every context makes its GPU's streams, one after the other: the runtime deals streams out
            // over a handful of hardware queues (4 by default), and two ACTIVE streams on one hardware queue wait for each
            // other's copies — `up` and `down` on one queue is H2D and D2H taking turns (28 GB/s each way; seen in a process
            // that already held streams: bench.py with torch, GPU_MAX_HW_QUEUES=4 against 16).  Which queue a stream gets
            // is the runtime's business (three streams created back to back shared one in a bare process, a stream per
            // slab on top of `up` / `down` did inside bench.py): separate_lane_streams() below measures and replaces.
            dpx_stream::Lane &ln = *s->lanes[k];
            e = hipStreamCreateWithFlags(&ln.up, hipStreamNonBlocking);
            if (e == hipSuccess && s->path == DPX_STREAM_PATH_STAGED) e = hipStreamCreateWithFlags(&ln.run, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&ln.down, hipStreamNonBlocking);
        }
        if (e == hipSuccess) {
            if (s->path == DPX_STREAM_PATH_STAGED) b.stream = s->lanes[k % (size_t)n_ctx]->run;
            else { e = hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking); b.owns_stream = e == hipSuccess; }
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b.ev_up, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b.ev_run, hipEventDisableTiming);
        // an event belongs to the device it is recorded on: `done` is recorded where the slab's output leaves — its own GPU,
        // or the ring's first GPU when the output is gathered there
        if (e == hipSuccess && s->gathered(k)) {           // where this slab's output lands on the ring's first GPU
            e = hipSetDevice(ctxs[0]->device);
            if (e == hipSuccess) e = hipMalloc(&b.g_out, s->slab_out + 16);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&b.ev_gather, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&b.done, hipEventDisableTiming);
            if (e == hipSuccess) e = hipSetDevice(b.ctx->device);
        } else if (e == hipSuccess) {
            e = hipEventCreateWithFlags(&b.done, hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            dpx_stream_destroy(s);
            return fail(DPX_ERR_HIP, "stream slab allocation failed: %s", hipGetErrorString(e));
        }
    }
    if (s->gather_rccl) {
        std::vector<int> devs;
        for (int i = 0; i < n_ctx; ++i) devs.push_back(ctxs[i]->device);
        s->comms.assign((size_t)n_ctx, nullptr);
        // RCCL greets on stdout when its first communicator is made ("RCCL version : ...", four lines) — and stdout is the
        // sample stream of the `doppler` command.  File descriptor 1 points at stderr while the communicators are made.
        fflush(stdout);
        const int saved_out = dup(1);
        if (saved_out >= 0) (void)dup2(2, 1);
        const int rc = rccl().CommInitAll(s->comms.data(), n_ctx, devs.data());
        fflush(stdout);
        if (saved_out >= 0) { (void)dup2(saved_out, 1); close(saved_out); }
        hipError_t e = hipSetDevice(ctxs[0]->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->gather_stream, hipStreamNonBlocking);
        if (rc != 0 || e != hipSuccess) {
            const std::string why = rc != 0 ? rccl().GetErrorString(rc) : hipGetErrorString(e);
            dpx_stream_destroy(s);
            return fail(DPX_ERR_HIP, "RCCL gather: ncclCommInitAll over %d device(s) failed: %s", n_ctx, why.c_str());
        }
    }
    if (s->path == DPX_STREAM_PATH_STAGED && !(o.path & DPX_STREAM_NO_PROBE))
        for (int i = 0; i < n_ctx; ++i) separate_lane_streams(s, (size_t)i);
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

int dpx_stream_create_multi(dpx_ctx *const *ctxs, int n_ctx, int in_fmt, int out_fmt, uint32_t samplerate,
                            uint32_t samplenum0, size_t slab_bytes, int slabs_per_ctx, dpx_stream **out)
{
    return dpx_stream_create_opts(ctxs, n_ctx, in_fmt, out_fmt, samplerate, samplenum0, slab_bytes, slabs_per_ctx, nullptr, out);
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
    for (dpx_ctx *c : s->ctxs) {                     // hipFree / hipHostFree wait for the device: a resident block kernel leaves first
        std::lock_guard<std::recursive_mutex> lock(c->dev->mu);
        (void)hipSetDevice(c->device);
        (void)resident_stop_device(c);
    }
    for (dpx_stream_slab &b : s->slabs) {
        if (!b.ctx) continue;
        std::lock_guard<std::recursive_mutex> lock(b.ctx->dev->mu);
        (void)hipSetDevice(b.ctx->device);
        if (b.stream) (void)hipStreamSynchronize(b.stream);
        if (b.done) (void)hipEventSynchronize(b.done);      // (recorded on the GPU's `down` stream when the output is staged)
        if (b.h_in) (void)hipHostFree(b.h_in);
        if (b.h_out) (void)hipHostFree(b.h_out);
        if (b.d_in) (void)hipFree(b.d_in);
        if (b.d_out) (void)hipFree(b.d_out);
        release(b.dev);
        if (b.done) (void)hipEventDestroy(b.done);
        if (b.ev_up) (void)hipEventDestroy(b.ev_up);
        if (b.ev_run) (void)hipEventDestroy(b.ev_run);
        if (b.ev_gather) { (void)hipEventSynchronize(b.ev_gather); (void)hipEventDestroy(b.ev_gather); }
        if (b.g_out) { (void)hipSetDevice(s->ctxs[0]->device); (void)hipFree(b.g_out); (void)hipSetDevice(b.ctx->device); }
        if (b.stream && b.owns_stream) (void)hipStreamDestroy(b.stream);
    }
    for (size_t i = 0; i < s->lanes.size(); ++i) {
        (void)hipSetDevice(s->ctxs[i]->device);
        if (s->lanes[i]->up) { (void)hipStreamSynchronize(s->lanes[i]->up); (void)hipStreamDestroy(s->lanes[i]->up); }
        if (s->lanes[i]->down) { (void)hipStreamSynchronize(s->lanes[i]->down); (void)hipStreamDestroy(s->lanes[i]->down); }
        if (s->lanes[i]->run) { (void)hipStreamSynchronize(s->lanes[i]->run); (void)hipStreamDestroy(s->lanes[i]->run); }
        for (hipStream_t st : s->lanes[i]->parked) (void)hipStreamDestroy(st);
    }
    if (s->gather_stream) {
        (void)hipSetDevice(s->ctxs[0]->device);
        (void)hipStreamSynchronize(s->gather_stream);
        (void)hipStreamDestroy(s->gather_stream);
    }
    for (void *c : s->comms)
        if (c) (void)rccl().CommDestroy(c);
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

}  // extern "C"

namespace {

// The staged path's three streams of one GPU must not share a hardware queue.  The runtime deals streams out over a few
// hardware queues (GPU_MAX_HW_QUEUES, 4 by default) by rules of its own, and what queues behind a copy in one stream — the
// event record that releases the next stage — holds up every other stream on that queue: `up` and `down` on one queue is
// H2D and D2H taking turns at the link (28 GB/s each way instead of 47: measured with the streams created back to back in
// a fresh process, and with a stream per slab inside bench.py).  So the ring measures once, when it is created: one stream
// is given two copies, another a 16-byte launch — on a queue of its own the launch is back in ~20 us, on a shared one only
// after the copies (`up` against `down`, `down` against `up`, each against `run`).  A stream that shares is parked — kept,
// idle, so that its replacement is dealt another queue — and replaced; at most eight rounds, ~2 ms each; the outcome is in
// dpx_stream_describe.  Performance only: a ring whose streams still share is slow, never wrong.
void separate_lane_streams(dpx_stream *s, size_t lane_index)
{
    dpx_stream::Lane &ln = *s->lanes[lane_index];
    dpx_stream_slab &b = s->slabs[lane_index];                  // the GPU's first slab lends its buffers
    if (hipSetDevice(b.ctx->device) != hipSuccess || !ln.up || !ln.run || !ln.down || !b.d_in || !b.d_out) return;
    const size_t in_b = s->slab_bytes, out_b = s->slab_out;
    const size_t nb = std::min<size_t>(std::min(in_b, out_b), (size_t)16 << 20) & ~(size_t)15;
    if (nb < ((size_t)1 << 20)) return;
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); };
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    for (hipEvent_t &e : ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            for (hipEvent_t made : ev) if (made) (void)hipEventDestroy(made);
            return;
        }
    auto up2 = [&] { for (int i = 0; i < 2; ++i) { (void)hipMemcpyAsync(b.d_in, b.h_in, nb, hipMemcpyHostToDevice, ln.up); (void)hipEventRecord(ev[0], ln.up); } };
    auto down2 = [&] { for (int i = 0; i < 2; ++i) { (void)hipMemcpyAsync(b.h_out, b.d_out, nb, hipMemcpyDeviceToHost, ln.down); (void)hipEventRecord(ev[1], ln.down); } };
    auto run1 = [&] { (void)dpx::launch_copy(b.d_in, static_cast<char *>(b.d_in) + 64, 16, ln.run); (void)hipEventRecord(ev[2], ln.run); };
    auto replace = [&](hipStream_t &st) {
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) return false;
        ln.parked.push_back(st);
        st = fresh;
        return true;
    };
    // how long a 16-byte launch on `probe` takes to come back while `busy` holds two copies: ~20 us on a queue of its own,
    // the copies' ~600 us behind them
    auto tiny = [&](hipStream_t st) { (void)dpx::launch_copy(b.d_in, static_cast<char *>(b.d_in) + 64, 16, st); (void)hipEventRecord(ev[2], st); };
    auto held_up_once = [&](bool busy_is_up, hipStream_t busy, hipStream_t probe) {
        const clk::time_point t = clk::now();
        if (busy_is_up) up2(); else down2();
        tiny(probe);
        (void)hipEventSynchronize(ev[2]);
        const double t_probe = secs(t);
        (void)hipStreamSynchronize(busy);
        const double t_busy = secs(t);
        (void)hipStreamSynchronize(probe);
        return t_probe > 0.5 * t_busy;
    };
    // (a late wake-up of this thread on a loaded host looks like a held-up launch: a positive has to repeat)
    auto held_up = [&](bool busy_is_up, hipStream_t busy, hipStream_t probe) {
        return held_up_once(busy_is_up, busy, probe) && held_up_once(busy_is_up, busy, probe);
    };
    const bool debug = getenv("DPX_STREAM_DEBUG") != nullptr;
    up2(); down2(); run1(); tiny(ln.up); tiny(ln.down);          // first use of the three streams, untimed
    (void)hipDeviceSynchronize();
    for (int round = 0; round < 8; ++round) {
        ln.probes = round + 1;
        const bool ud = held_up(true, ln.up, ln.down), du = held_up(false, ln.down, ln.up);
        const bool ur = held_up(true, ln.up, ln.run), dr = held_up(false, ln.down, ln.run);
        ln.shared_queue = ud || du || ur || dr;
        if (debug) fprintf(stderr, "dpx_stream: lane %zu round %d: up holds down %d, down holds up %d, up holds run %d, down holds run %d\n",
                           lane_index, round, ud, du, ur, dr);
        if (!ln.shared_queue) break;
        bool ok = true;
        if (ur || dr) ok = replace(ln.run);
        if ((ud || du) && ok) ok = replace(ln.down);
        if (!ok) break;
        run1(); down2(); tiny(ln.down);
        (void)hipDeviceSynchronize();
    }
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    for (size_t k = lane_index; k < s->slabs.size(); k += s->lanes.size()) s->slabs[k].stream = ln.run;
}

// Hands queued D2H copies of one GPU to the runtime, oldest first, each only once its predecessor has finished (`upto` >= 0:
// everything up to and including that slab regardless — dpx_stream_next(k), which cannot wait for a copy nobody has issued).
// The caller holds the device's lock and has made the device current.
int pump_down(dpx_stream *s, dpx_stream::Lane &lane, long upto)
{
    std::lock_guard<std::mutex> lk(lane.mu);
    while (!lane.down_q.empty()) {
        const size_t k = lane.down_q.front();
        bool forced = false;
        if (upto >= 0)
            for (size_t q : lane.down_q) forced = forced || q == (size_t)upto;
        if (!forced && lane.last_down >= 0 && hipEventQuery(s->slabs[(size_t)lane.last_down].done) != hipSuccess) break;
        dpx_stream_slab &b = s->slabs[k];
        const bool via_gpu0 = s->gathered(k);                      // its output was sent to the ring's first GPU (RCCL gather)
        DPX_HIP(hipStreamWaitEvent(lane.down, via_gpu0 ? b.ev_gather : b.ev_run, 0));
        DPX_HIP(hipMemcpyAsync(b.h_out, via_gpu0 ? b.g_out : b.d_out, b.out_bytes, hipMemcpyDeviceToHost, lane.down));
        DPX_HIP(hipEventRecord(b.done, lane.down));
        lane.last_down = (long)k;
        lane.down_q.pop_front();
    }
    return DPX_OK;
}

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
    const void *k_in = s->in_direct() ? b.m_in : b.d_in;
    void *k_out = s->out_direct() ? b.m_out : b.d_out;
    const size_t k_slab = (size_t)(&b - s->slabs.data());
    dpx_stream::Lane &lane = *s->lanes[k_slab % s->lanes.size()];
    hipStream_t st_up = s->per_slab() ? b.stream : lane.up, st_down = s->per_slab() ? b.stream : lane.down;
    if (!s->in_direct()) {
        DPX_HIP(hipMemcpyAsync(b.d_in, b.h_in, b.job_in_bytes, hipMemcpyHostToDevice, st_up));
        if (st_up != b.stream) {
            DPX_HIP(hipEventRecord(b.ev_up, st_up));
            DPX_HIP(hipStreamWaitEvent(b.stream, b.ev_up, 0));
        }
    }
    if (!s->copy_only) {
        rc = run_plan(b.plan, b.dev, k_in, s->in_fmt, k_out, s->out_fmt, b.job_fma, b.job_geom, b.stream);
        if (rc != DPX_OK) return rc;
    } else if (s->in_direct() || s->out_direct()) {       // calibration: the kernel's side(s) of the link with no arithmetic
        const size_t nb = (b.job_in_bytes < b.out_bytes ? b.job_in_bytes : b.out_bytes) & ~(size_t)15;
        if (nb && dpx::launch_copy(k_in, k_out, nb, b.stream) != DPX_OK) return fail(DPX_ERR_HIP, "copy launch failed");
    }
    if (!s->out_direct()) {
        if (st_down != b.stream) {
            DPX_HIP(hipEventRecord(b.ev_run, b.stream));
            if (s->gathered(k_slab)) {
                // RCCL ordered gather: this GPU sends, the ring's first GPU receives — one group, so that the pair is
                // issued together whichever thread gets here first; the D2H then queues on GPU 0's `down` stream and is
                // handed to the runtime by dpx_stream_next / that GPU's own submits (paced like its own slabs)
                const Rccl &nc = rccl();
                const size_t g = k_slab % s->lanes.size();
                std::lock_guard<std::mutex> glk(s->gather_mu);
                DPX_NCCL(nc.GroupStart());
                int e1 = nc.Send(b.d_out, b.out_bytes, kNcclUint8, 0, s->comms[g], b.stream);
                (void)hipSetDevice(s->ctxs[0]->device);
                int e2 = nc.Recv(b.g_out, b.out_bytes, kNcclUint8, (int)g, s->comms[0], g == 0 ? b.stream : s->gather_stream);
                const int e3 = nc.GroupEnd();
                hipError_t he = hipEventRecord(b.ev_gather, g == 0 ? b.stream : s->gather_stream);
                (void)hipSetDevice(b.ctx->device);
                if (e1 || e2 || e3) return fail(DPX_ERR_HIP, "RCCL gather of a slab failed: %s", nc.GetErrorString(e1 ? e1 : e2 ? e2 : e3));
                if (he != hipSuccess) return fail(DPX_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(he));
                ++s->gathered_slabs;
                s->gathered_bytes += b.out_bytes;
                dpx_stream::Lane &l0 = *s->lanes[0];
                std::lock_guard<std::mutex> lk(l0.mu);
                l0.down_q.push_back(k_slab);
            } else {
                {
                    std::lock_guard<std::mutex> lk(lane.mu);
                    lane.down_q.push_back(k_slab);
                }
                rc = pump_down(s, lane, s->paced ? -1 : (long)k_slab);
                if (rc != DPX_OK) return rc;
            }
        } else {
            DPX_HIP(hipMemcpyAsync(b.h_out, b.d_out, b.out_bytes, hipMemcpyDeviceToHost, st_down));
            DPX_HIP(hipEventRecord(b.done, st_down));
        }
    } else {
        DPX_HIP(hipEventRecord(b.done, b.stream));
    }
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

extern "C" {

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
        dpx::finalize(b.plan, g.tile(), ctx->choice, ctx->tuning);
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
    const bool staged_out = !s->out_direct() && !s->per_slab();
    const size_t dl = s->down_lane(s->tail);         // the GPU whose link this output leaves through (RCCL gather: the first)
    if (staged_out) {                      // the slab's D2H may still be queued behind its GPU's previous one (which is done: it was handed out)
        DPX_ENTER(s->ctxs[dl]);
        const int rc = pump_down(s, *s->lanes[dl], (long)s->tail);
        if (rc != DPX_OK) return rc;
    }
    DPX_HIP(hipEventSynchronize(b.done));
    if (staged_out) {                      // ... and the next one of this GPU can go now
        DPX_ENTER(s->ctxs[dl]);
        const int rc = pump_down(s, *s->lanes[dl], -1);
        if (rc != DPX_OK) return rc;
    }
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
    std::lock_guard<std::mutex> lk(s->enq_mu);          // the enqueue threads add their parts under this lock
    *out = s->stats;
    return DPX_OK;
}

int dpx_stream_describe(const dpx_stream *s, uint32_t *path, int *numa_nodes, size_t cap, size_t *n_slabs)
{
    if (!s) return fail(DPX_ERR_ARG, "bad argument");
    if (path) {
        *path = s->path | (s->copy_only ? DPX_STREAM_COPY_ONLY : 0u) | (s->paced ? 0u : DPX_STREAM_UNPACED);
        int probes = 0;
        bool shared = false;
        for (const auto &ln : s->lanes) { probes = std::max(probes, ln->probes); shared = shared || ln->shared_queue; }
        *path |= ((uint32_t)probes & 0xfu) << 16 | (shared ? DPX_STREAM_SHARED_QUEUE : 0u) | (s->gather_rccl ? DPX_STREAM_DESCRIBE_RCCL : 0u);
    }
    if (n_slabs) *n_slabs = s->slabs.size();
    for (size_t k = 0; numa_nodes && k < cap && k < s->slabs.size(); ++k) numa_nodes[k] = s->slabs[k].numa_node;
    return DPX_OK;
}

int dpx_stream_samplenum(const dpx_stream *s, uint32_t *samplenum)
{
    if (!s || !samplenum) return fail(DPX_ERR_ARG, "bad argument");
    *samplenum = s->samplenum;
    return DPX_OK;
}

}  // extern "C"
