/*
 * doppler_hip_debug.h — measurement knobs, planner self-checks and helpers of libdoppler_hip.so.
 *
 * Nothing here is needed to bind or to use the hot path (doppler_hip.h); none of it changes a result.  What it is for:
 *   dpx_options / dpx_set_options / dpx_set_tuning   kernel-shape alternatives timed against each other (tools/ab.py,
 *                                                    profiles/): which launch shape produces the same bytes;
 *   dpx_plan_describe / _layout / _simulate          the planner's stretch list, launch layout and a host mirror of the
 *                                                    kernels' index arithmetic (tests/test_host_logic.py, no device needed);
 *   dpx_debug_copy                                   the memory system's own copy rate (calibration of profiles/);
 *   dpx_stream_get_stats / _pending, dpx_plan_n_samples, dpx_set_resident / dpx_resident_stats / _info   introspection, A/B;
 *   dpx_malloc ... dpx_synchronize                   device memory for callers without a HIP binding (ctypes tests, the CLI).
 */
#ifndef DOPPLER_HIP_DEBUG_H
#define DOPPLER_HIP_DEBUG_H

#include "doppler_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-shape knobs for measurements (all 0 = the planner's own choice).  They never change a result,
 * only which launch shape produces it; profiles/ names the values behind every alternative it reports. */
typedef struct dpx_options {
    uint32_t rows_mult;      /* rows kernel: row length = rows_mult * lcm(period, 4) samples */
    uint32_t rows_maxl;      /* rows kernel: longest row considered */
    uint32_t rows_r;         /* rows kernel: rows per wavefront (2 or 4) */
    uint32_t rows_compute;   /* rows kernel: for periods of at least this many samples a launch leaves its table alone, every wavefront
                              * evaluates its columns' correctors once for its rows (0 = the planner's threshold, 2049;
                              * 0xffffffff = never; 1 = always) */
    uint32_t walk_waves;     /* span kernel: wavefronts per workgroup (2, 4, 5 or 8) */
    uint32_t walk_span;      /* span kernel: most rows of a matrix one workgroup keeps its column window for.  0 = the planner's
                              * own cut: spans of 8, a matrix of up to 12 rows whole, several adjacent windows per workgroup for
                              * spans of up to 4 rows; >= 2: spans of at most
                              * that many rows, one window per workgroup, two rows per wavefront per turn */
    uint32_t walk_flags;     /* bit 0: a span launch of ONE matrix reads its descriptors from memory like a many-matrix launch instead
                              * of taking the matrix from its kernel arguments; bits 8..: row-length target in KiSamples */
    uint32_t sub_lg;         /* span launches over long streams are dealt out as sub-launches of about 2^sub_lg samples, back
                              * to back on the stream (0 = the default, 28: 1 GiB of i16; >= 48 = one launch whatever the length) */
    uint64_t walk_tilemin;   /* span plans: uncovered gaps at least this long (samples) get a tile-kernel launch */
} dpx_options;
/* applies to plans created afterwards; NULL restores the defaults */
int dpx_set_options(dpx_ctx *ctx, const dpx_options *opt);

/* One stretch of the stream in which the counter is a closed form of the
 * sample index j (relative to `first`):  period == 0: n = n_start + j;
 * period > 0: n = ((n_start - 1 + j) mod period) + 1.  lut_len > 0: one period of
 * correctors is tabulated in device memory at plan time for this stretch
 * (lut_len == 0: the kernels evaluate sincos themselves). */
typedef struct {
    uint64_t first, count;
    float ratio;
    uint32_t n_start, period, lut_len;
} dpx_stretch;

/* Host-only (no device needed): the stretch list the planner derives for a
 * segment list; writes at most `cap` entries, *n_out = total number. */
int dpx_plan_describe(const dpx_segment *segs, size_t n_segs, uint32_t samplerate,
                      uint32_t samplenum0, int variant, dpx_stretch *out, size_t cap,
                      size_t *n_out, uint32_t *final_samplenum);

/* Host-only self-check of a plan's launch list (no device needed): for every sample,
 * the counter value the kernels' index arithmetic selects (counters[n_samples]) and
 * how many launches write it (writes[n_samples], must be exactly 1 everywhere).
 * block / vecs / variant as in dpx_set_tuning (0 = defaults); in_fmt / out_fmt: the format pair of the launch being
 * mirrored (a span launch cuts its grid per pair). */
int dpx_plan_simulate(const dpx_segment *segs, size_t n_segs, uint32_t samplerate,
                      uint32_t samplenum0, int block, int vecs, int variant, const struct dpx_options *opt,
                      int in_fmt, int out_fmt, uint32_t *counters, uint8_t *writes, uint64_t n_samples);

/* Host-only: how the planner lays a segment list out over the kernels (what dpx_run_device will launch). */
typedef struct dpx_layout {
    uint64_t n_samples;
    uint64_t rows_samples;      /* produced by rows-kernel matrices */
    uint64_t walk_samples;      /* produced by span-kernel matrices */
    uint64_t tile_samples;      /* inside tile-kernel launches */
    uint64_t single_samples;    /* evaluated one by one (ragged edges of rows launches, leftover ranges) */
    uint64_t table_entries;     /* (cos, sin) pairs tabulated at plan time */
    uint32_t n_stretches;
    uint32_t rows_launches, tile_launches, walk_launches;
    uint32_t walk_matrices, walk_workgroups, leftover_ranges, leftover_workgroups;
    uint32_t f32_i16_by_tiles;  /* 1: an f32 -> i16 or i16 -> f32 run of this plan is ONE tile-kernel launch over the whole stream instead of
                                 * the launches counted above (plans of many matrices: csrc/dpx_planner.h, launches_for) */
    uint32_t reserved;
} dpx_layout;
int dpx_plan_layout(const dpx_segment *segs, size_t n_segs, uint32_t samplerate, uint32_t samplenum0,
                    int block, int vecs, int variant, const struct dpx_options *opt, dpx_layout *out);

int dpx_plan_n_samples(const dpx_plan *plan, uint64_t *n_samples);
/* where dpx_plan_segments / dpx_plan_const spent its time, microseconds: [0] stretch list (the counter rule's closed form per
 * segment), [1] launch layout (kernel choice, spans, hint tables), [2] device image: build, upload, table kernels, the wait */
int dpx_plan_timing(const dpx_plan *plan, double out_us[3]);

/* The resident block kernel behind dpx_shift_block / dpx_shift_block_async (on by default; DPX_RESIDENT=0 in the environment
 * turns it off for every context): on = 0 sends every block through a launch of its own again (round 3's path, kept as the
 * fallback), for A/B timing.  dpx_resident_stats: kernel launches and blocks served through doorbells so far. */
int dpx_set_resident(dpx_ctx *ctx, int on);
int dpx_resident_stats(const dpx_ctx *ctx, uint64_t *launches, uint64_t *blocks);
/* The same with how the launches ended.  Every launch of the resident kernel ends in exactly one of two ways — it was asked
 * to leave (`stops`: any other work of the context, another format pair, dpx_set_resident(0), dpx_ctx_destroy) or it was
 * found parked (`idle_exits`: 2 ms without a block, or a ticket meant for another instance of the kernel) — so
 *     launches == stops + idle_exits + running
 * holds whenever no call is in progress, on an idle box and on a loaded one (tests assert this, not launch counts). */
typedef struct dpx_resident_counters {
    uint64_t launches, blocks, stops, idle_exits;
    uint32_t running;            /* a kernel has been launched and not yet been seen parked */
    uint32_t tickets_in_flight;  /* dpx_shift_block_async tickets not yet waited for */
    uint32_t slots_parked;       /* staging slots whose workgroup is not polling */
    uint32_t reserved;
} dpx_resident_counters;
int dpx_resident_info(dpx_ctx *ctx, dpx_resident_counters *out);

int dpx_stream_pending(const dpx_stream *s, int *n_in_flight);
/* Host time dpx_stream_submit has spent so far, by part (microseconds, summed over `slabs` calls): planning (stretch list +
 * launch layout; skipped when the slab buffer's resident plan was made from the same segments at the same counter:
 * `plans_reused`), building and uploading the plan's device image, and enqueueing the two copies, the launch and the event. */
typedef struct dpx_stream_stats {
    uint64_t slabs, plans_reused;
    double plan_us, upload_us, enqueue_us, total_us;
} dpx_stream_stats;
int dpx_stream_get_stats(const dpx_stream *s, dpx_stream_stats *out);

/* How the slabs of a ring cross PCIe — for the same-process A/B behind the defaults (profiles/r06_ring.md); never changes a byte.
 *   DPX_STREAM_PATH_DEFAULT     by slab size, the path that measured fastest: DIRECT below 1 MiB, DIRECT_OUT below 4 MiB, STAGED from there
 *   DPX_STREAM_PATH_DIRECT      the fused kernel loads from the pinned input slab and stores to the pinned output slab
 *                               (host-mapped memory): no HBM staging, no copy engine, one launch per slab (13 us per 8 KiB slab)
 *   DPX_STREAM_PATH_STAGED      copy engine H2D -> kernel HBM to HBM -> copy engine D2H: every H2D of a GPU on its `up` stream, every
 *                               D2H on `down` (handed to the runtime one at a time: DPX_STREAM_UNPACED queues them all at submit),
 *                               launches on `run`; the three streams are probed for a shared hardware queue when the ring is made
 *                               (DPX_STREAM_NO_PROBE skips that).  45.6-47.5 GB/s each way at 16-64 MiB slabs: 0.98 of the link
 *   DPX_STREAM_PATH_DIRECT_IN   kernel loads from the host slab, output staged;  _DIRECT_OUT: input staged, kernel stores to host
 *   DPX_STREAM_PATH_STAGED_PER_SLAB   rounds 2-5: H2D -> kernel -> D2H on one stream per slab (28-33 GB/s: the copies take turns)
 *   | DPX_STREAM_COPY_ONLY      calibration: the same slabs on the same path without arithmetic (a plain copy kernel on the
 *                               kernel's side of the link, nothing at all between the two engine copies of STAGED)
 * in/out_host_flags: extra hipHostMalloc flags of the input / output slabs (hipHostMallocNonCoherent 0x80000000,
 * hipHostMallocWriteCombined 0x4, hipHostMallocNumaUser 0x20000000).
 * opt == NULL: the defaults (path from DPX_STREAM_PATH, gather from DPX_STREAM_GATHER in the environment when set). */
#define DPX_STREAM_PATH_DEFAULT 0u
#define DPX_STREAM_PATH_DIRECT 1u
#define DPX_STREAM_PATH_STAGED 2u
#define DPX_STREAM_PATH_DIRECT_IN 3u
#define DPX_STREAM_PATH_DIRECT_OUT 4u
#define DPX_STREAM_PATH_STAGED_PER_SLAB 5u
#define DPX_STREAM_COPY_ONLY 0x100u
#define DPX_STREAM_UNPACED 0x200u
#define DPX_STREAM_NO_PROBE 0x400u        /* STAGED: take the three streams of a GPU as the runtime deals them (A/B) */
#define DPX_STREAM_SHARED_QUEUE 0x800u    /* dpx_stream_describe only: after the last probe round two of a GPU's streams still shared a hardware queue */
#define DPX_STREAM_DESCRIBE_RCCL 0x1000u  /* dpx_stream_describe only: the ring gathers through RCCL */
/* gather — how the outputs of a ring over several GPUs reach the host:
 *   DPX_STREAM_GATHER_D2H   every GPU copies its own slabs to their pinned output buffers over its own PCIe link   [the default]
 *   DPX_STREAM_GATHER_RCCL  BASELINE.json's north_star form: the slabs of GPUs 1..N-1 go over xGMI into the first GPU
 *                           (one ncclSend / ncclRecv pair per slab, single-process communicators from ncclCommInitAll;
 *                           librccl is loaded on demand) and leave from there — every output byte through ONE PCIe link.
 *                           Needs distinct devices; runs on the staged path.  `doppler --gather rccl`, or DPX_STREAM_GATHER=rccl
 *                           in the environment for rings created without options.
 *   | DPX_STREAM_GATHER_SELF  (tests on a one-GPU box) the first GPU's own slabs take the same road: a send / recv to itself */
#define DPX_STREAM_GATHER_D2H 0u
#define DPX_STREAM_GATHER_RCCL 1u
#define DPX_STREAM_GATHER_SELF 0x100u
typedef struct dpx_stream_options {
    uint32_t path;
    uint32_t in_host_flags, out_host_flags;
    uint32_t gather;
} dpx_stream_options;
int dpx_stream_create_opts(dpx_ctx *const *ctxs, int n_ctx, int in_fmt, int out_fmt, uint32_t samplerate,
                           uint32_t samplenum0, size_t slab_bytes, int slabs_per_ctx, const dpx_stream_options *opt,
                           dpx_stream **stream);
/* the path a ring runs on (one of the above | its flags | probe rounds of the staged path's streams << 16) and the NUMA node each slab's pinned buffers were placed on (-1: the caller's policy) */
int dpx_stream_describe(const dpx_stream *s, uint32_t *path, int *numa_nodes, size_t cap, size_t *n_slabs);

/* Same access pattern, no arithmetic: 16-byte non-temporal copy of n_bytes.
 * Calibration only (profiles/: what the memory system gives a pure stream). */
int dpx_debug_copy(dpx_ctx *ctx, const void *d_in, void *d_out, size_t n_bytes, void *hip_stream);

/* Measurement knobs (0 keeps the current value); they apply to plans created afterwards.
 * block / vecs: tile-kernel geometry, lanes per workgroup and 4-sample groups per lane: 256 x 1, 128 x 2 or 64 x 4 (the same
 *          1024-sample tile; the three shapes that are built).  Until a call names one, every launch picks between them
 *          from its output format and whether the plan has tile tables (measured: DESIGN.md section 4);
 *          block = vecs = -1 returns to that.
 * variant: 3 = auto: rows kernel (one period of correctors tabulated) for up to eight long stretches (const mode),
 *              span kernel for more (track mode) and for odd periods, tile kernel for the rest;
 *          1 = sincos per sample wherever the period allows it (>= 4);
 *          2 = tabulate whenever the period fits;
 *          4 = auto, but keep everything on the tile kernel;
 *          5 = auto, but use the span kernel wherever a stretch qualifies;
 *          6 = auto, but never the span kernel. */
int dpx_set_tuning(dpx_ctx *ctx, int block, int vecs, int variant);

/* --------------------------------------------- device memory helpers
 * For callers without their own HIP runtime binding (ctypes tests, the CLI). */
int dpx_malloc(dpx_ctx *ctx, size_t bytes, void **d_ptr);
int dpx_free(dpx_ctx *ctx, void *d_ptr);
int dpx_memcpy_h2d(dpx_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int dpx_memcpy_d2h(dpx_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int dpx_synchronize(dpx_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
